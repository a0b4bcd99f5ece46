# round-6 measurement batch (run on the GPU box through gpurun): bench lines, kernel traces, PMC traffic, SQ counters (JSON + txt) of the
# persistent sweep (config 3) and of the exact latent draws, config 5 in both latent modes. Outputs under gpurun_out/${T}_*.
set -x
cd $GRAFT_REPO_ROOT
T=${1:-r06_m}
R=$GRAFT_REPO_ROOT
python bench.py > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
python bench.py --gpus 1 --steps 20 --warmup 5 --no-other-configs > gpurun_out/${T}_bench_driver_cmd.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
Q="--cpu-seconds 0 --fit-iters 0 --no-other-configs --long-seconds 0"
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${T}_trace -- python $R/bench.py --steps 10 --warmup 2 $Q > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/${T}_pmc_fetch -o p -- python $R/bench.py --steps 3 --warmup 1 $Q --no-kernel-timing > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/${T}_pmc_write -o p -- python $R/bench.py --steps 3 --warmup 1 $Q --no-kernel-timing > /dev/null 2>&1
cd $R
python scripts/pmc_to_traffic.py gpurun_out/${T}_pmc_fetch gpurun_out/${T}_pmc_write gpurun_out/${T}_pmc_traffic | head -12
python scripts/rocpd_summary.py $(ls gpurun_out/${T}_trace/*/*_results.db | head -1) gpurun_out/${T}_kernel_stats.txt "rocprofv3 --kernel-trace --stats -- python bench.py --steps 10 --warmup 2 $Q" | head -8
rm -rf gpurun_out/${T}_trace gpurun_out/${T}_pmc_fetch gpurun_out/${T}_pmc_write
# SQ counters of the persistent sweep at config 3: txt + the JSON bench.py's roofline.second_bound reads
SQ_BENCH_ARGS="--steps 3 --warmup 1 --long-seconds 0 --no-other-configs" SQ_KERNEL=k_mf_resident SQ_KEEP=1 bash scripts/prof_sq.sh ${T}_sq3 > gpurun_out/${T}_sq_counters_config3.txt 2>&1
python scripts/sq_to_json.py gpurun_out/${T}_sq3 k_mf_resident sweep_V_resident gpurun_out/r06_sq_counters_config3.json "profiles/${T}_sq_counters_config3.txt (rocprofv3 --pmc, bench.py --steps 3 --warmup 1)" > /dev/null
rm -rf gpurun_out/${T}_sq3_a gpurun_out/${T}_sq3_b gpurun_out/${T}_sq3_c
# config 5 at full size: kernel trace in both latent modes, PMC traffic, SQ counters of the latent kernels
bash scripts/prof_cfg.sh ${T}_cfg5 --config 5 --scale 1.0 --steps 2 --warmup 1 > /dev/null
bash scripts/prof_cfg.sh ${T}_cfg5_philox --config 5 --scale 1.0 --latent philox --steps 2 --warmup 1 > /dev/null
bash scripts/prof_cfg.sh ${T}_cfg3_classification --config 3 --task classification --steps 10 --warmup 2 > /dev/null
bash scripts/prof_cfg.sh ${T}_cfg4 --config 4 --steps 20 --warmup 3 > /dev/null
C5="--config 5 --scale 1.0 --steps 2 --warmup 1 --cpu-iters 0 --fit-iters 0 --no-kernel-timing --long-seconds 0"
cd /tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/${T}_c5_fetch -o p -- python $R/bench.py $C5 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/${T}_c5_write -o p -- python $R/bench.py $C5 > /dev/null 2>&1
cd $R
python scripts/pmc_to_traffic.py gpurun_out/${T}_c5_fetch gpurun_out/${T}_c5_write gpurun_out/${T}_pmc_traffic_config5 | head -40
rm -rf gpurun_out/${T}_c5_fetch gpurun_out/${T}_c5_write
SQ_BENCH_ARGS="--config 5 --scale 1.0 --steps 1 --warmup 1 --long-seconds 0" SQ_KERNEL=k_lat bash scripts/prof_sq.sh ${T}_sql > gpurun_out/${T}_lat_sq_counters.txt 2>&1
rm -rf gpurun_out/${T}_sql_a gpurun_out/${T}_sql_b gpurun_out/${T}_sql_c
MFM_LATENT_TIMING=1 python bench.py --config 5 --scale 1.0 --steps 4 --warmup 1 --cpu-seconds 0 --fit-iters 0 --long-seconds 0 2>&1 >/dev/null | grep "^\[latent\]" | tail -3 > gpurun_out/${T}_latent_timing_config5.txt
MFM_LATENT_TIMING=1 python bench.py --config 3 --task classification --steps 10 --warmup 2 --cpu-seconds 0 --fit-iters 0 --long-seconds 0 --no-other-configs 2>&1 >/dev/null | grep "^\[latent\]" | tail -3 > gpurun_out/${T}_latent_timing_config3_classification.txt
python scripts/bench_predict.py > gpurun_out/${T}_predict.txt 2>&1
for f in gpurun_out/${T}_bench*.json; do python - $f <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
cb = d.get("cpu_baseline") or {}
print("%-56s value %9.3f  ms/step %9.3f  cpu %s  fit %s  setup_s %s" % (sys.argv[1], d["value"], d["ms_per_step"], cb.get("value"), (d.get("fit") or {}).get("fit_it_per_s"), d["config"].get("setup_s")))
for k, v in (d.get("other_configs") or {}).items():
    print("    %-60s %s it/s  (cpu %s)  latent %s" % (k, v.get("it_per_s"), v.get("cpu_it_per_s"), v.get("latent")))
PY
done
