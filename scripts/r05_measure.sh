# round-5 measurement batch (run on the GPU box through gpurun): bench line, kernel traces, PMC traffic (config 3 and config 5), SQ
# counters of the streamed chain, its phase stamps and per-step timeline, the other workloads. Outputs under gpurun_out/${T}_*.
set -x
cd $GRAFT_REPO_ROOT
T=${1:-r05_m}
R=$GRAFT_REPO_ROOT
python bench.py > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
cd /tmp && export TMPDIR=/tmp
Q="--cpu-seconds 0 --fit-iters 0 --no-other-configs --long-seconds 0"
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${T}_trace -- python $R/bench.py --steps 10 --warmup 2 $Q > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/${T}_pmc_fetch -o p -- python $R/bench.py --steps 3 --warmup 1 $Q --no-kernel-timing > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/${T}_pmc_write -o p -- python $R/bench.py --steps 3 --warmup 1 $Q --no-kernel-timing > /dev/null 2>&1
cd $R
python scripts/pmc_to_traffic.py gpurun_out/${T}_pmc_fetch gpurun_out/${T}_pmc_write gpurun_out/${T}_pmc_traffic | head -12
python scripts/rocpd_summary.py $(ls gpurun_out/${T}_trace/*/*_results.db | head -1) gpurun_out/${T}_kernel_stats.txt "rocprofv3 --kernel-trace --stats -- python bench.py --steps 10 --warmup 2 $Q" | head -8
rm -rf gpurun_out/${T}_trace gpurun_out/${T}_pmc_fetch gpurun_out/${T}_pmc_write
# config 5 at full size: kernel trace, PMC traffic, SQ counters, the streamed chain's phases and timeline
C5="--config 5 --scale 1.0 --steps 2 --warmup 1 --cpu-iters 0 --fit-iters 0 --no-kernel-timing --long-seconds 0"
bash scripts/prof_cfg.sh ${T}_cfg5 --config 5 --scale 1.0 --steps 2 --warmup 1 > /dev/null
bash scripts/prof_cfg.sh ${T}_cfg4 --config 4 --steps 20 --warmup 3 > /dev/null
cd /tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/${T}_c5_fetch -o p -- python $R/bench.py $C5 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/${T}_c5_write -o p -- python $R/bench.py $C5 > /dev/null 2>&1
cd $R
python scripts/pmc_to_traffic.py gpurun_out/${T}_c5_fetch gpurun_out/${T}_c5_write gpurun_out/${T}_pmc_traffic_config5 | head -30
rm -rf gpurun_out/${T}_c5_fetch gpurun_out/${T}_c5_write
SQ_BENCH_ARGS="--config 5 --scale 1.0 --steps 1 --warmup 1 --long-seconds 0" SQ_KERNEL=k_cs_stream bash scripts/prof_sq.sh ${T}_sq5 > gpurun_out/${T}_sq_counters_config5.txt 2>&1
rm -rf gpurun_out/${T}_sq5_a gpurun_out/${T}_sq5_b gpurun_out/${T}_sq5_c
MFM_CS_TRACE_LAUNCH=40 MFM_CS_TRACE=$R/gpurun_out/${T}_cs_trace_item.txt MFM_CB_PROF=130 python bench.py $C5 > /dev/null 2> gpurun_out/${T}_cs_phases.txt
MFM_CS_TRACE_LAUNCH=41 MFM_CS_TRACE=$R/gpurun_out/${T}_cs_trace_user.txt MFM_CB_PROF=130 MFM_SETUP_TIMING=1 python bench.py $C5 > /dev/null 2> gpurun_out/${T}_cs_phases_user.txt
(grep "k_cs_stream\|streamed chain" gpurun_out/${T}_cs_phases_user.txt | tail -4; python scripts/cs_trace_analyze.py gpurun_out/${T}_cs_trace_item.txt 100 6; python scripts/cs_trace_analyze.py gpurun_out/${T}_cs_trace_user.txt 200 6) > gpurun_out/${T}_cs_stream_timeline.txt 2>&1
rm -f gpurun_out/${T}_cs_trace_item.txt gpurun_out/${T}_cs_trace_user.txt
MFM_NO_CB_STREAM=1 python bench.py --config 5 --scale 1.0 --steps 4 --warmup 1 --cpu-seconds 0 --fit-iters 0 --long-seconds 0 2>/dev/null | tail -1 > gpurun_out/${T}_bench_config5_full_batched_chain.json
python bench.py --config 2 --steps 200 --warmup 10 --cpu-seconds 10 2>/dev/null | tail -1 > gpurun_out/${T}_bench_config2.json
python bench.py --config 4 --steps 100 --warmup 5 --cpu-seconds 30 2>/dev/null | tail -1 > gpurun_out/${T}_bench_config4.json
python bench.py --config 5 --scale 1.0 --steps 6 --warmup 2 --cpu-seconds 0 --fit-iters 0 2>/dev/null | tail -1 > gpurun_out/${T}_bench_config5_full.json
python bench.py --config 5 --scale 0.1 --steps 10 --warmup 2 --cpu-seconds 20 --fit-iters 0 2>/dev/null | tail -1 > gpurun_out/${T}_bench_config5_scale0.1.json
python scripts/bench_predict.py > gpurun_out/${T}_predict.txt 2>&1
bash scripts/r05_overflow_bench.sh > gpurun_out/${T}_overflow_rows_scaling.txt 2>&1
for f in gpurun_out/${T}_bench*.json; do python - $f <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
cb = d.get("cpu_baseline") or {}
print("%-56s value %9.3f  ms/step %9.3f  cpu %s  fit %s  setup_s %s" % (sys.argv[1], d["value"], d["ms_per_step"], cb.get("value"), (d.get("fit") or {}).get("fit_it_per_s"), d["config"].get("setup_s")))
PY
done
