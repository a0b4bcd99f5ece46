# measurement batch (run on the GPU box through gpurun): bench line, kernel trace, PMC traffic, SQ counters,
# predictor timing, other workloads. Outputs under gpurun_out/r02_z_*; summaries are copied to profiles/ afterwards.
set -x
cd $GRAFT_REPO_ROOT
T=${1:-r03_a}
python bench.py > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/${T}_trace -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --cpu-seconds 0 --fit-iters 0 --no-other-configs > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${T}_pmc_fetch -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --cpu-seconds 0 --fit-iters 0 --no-kernel-timing --no-other-configs > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${T}_pmc_write -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --cpu-seconds 0 --fit-iters 0 --no-kernel-timing --no-other-configs > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python scripts/pmc_to_traffic.py gpurun_out/${T}_pmc_fetch gpurun_out/${T}_pmc_write gpurun_out/${T}_pmc_traffic | head -30
python scripts/rocpd_summary.py $(ls gpurun_out/${T}_trace/*/*_results.db | head -1) gpurun_out/${T}_kernel_stats.txt "rocprofv3 --kernel-trace --stats -- python bench.py --steps 10 --warmup 2 --cpu-seconds 0 --fit-iters 0 --no-other-configs" | head -14
rm -rf gpurun_out/${T}_trace
rm -rf gpurun_out/${T}_pmc_fetch gpurun_out/${T}_pmc_write/p_agent_info.csv
python scripts/bench_predict.py > gpurun_out/${T}_predict.txt 2>&1; tail -5 gpurun_out/${T}_predict.txt
# the other BASELINE workloads (parity-test shapes; reported next to their CPU-oracle rates)
python bench.py --config 2 --steps 200 --warmup 10 --cpu-seconds 10 2>/dev/null | tail -1 > gpurun_out/${T}_bench_config2.json
python bench.py --config 4 --steps 100 --warmup 5 --cpu-seconds 30 2>/dev/null | tail -1 > gpurun_out/${T}_bench_config4.json
python bench.py --config 5 --scale 0.1 --steps 10 --warmup 2 --cpu-seconds 30 --fit-iters 0 2>/dev/null | tail -1 > gpurun_out/${T}_bench_config5_scale0.1.json
python bench.py --config 5 --scale 1.0 --steps 5 --warmup 1 --cpu-seconds 0 --fit-iters 0 2>/dev/null | tail -1 > gpurun_out/${T}_bench_config5_full.json
bash scripts/prof_cfg.sh ${T}_cfg4 --config 4 --steps 20 --warmup 3 > /dev/null
bash scripts/prof_cfg.sh ${T}_cfg5 --config 5 --scale 1.0 --steps 2 --warmup 1 > /dev/null
for f in gpurun_out/${T}_bench*.json; do python - $f <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
cb = d.get("cpu_baseline") or {}
print("%-48s value %9.3f  ms/step %9.3f  cpu %s  fit %s  setup_s %s" % (sys.argv[1], d["value"], d["ms_per_step"], cb.get("value"), (d.get("fit") or {}).get("fit_it_per_s"), d["config"].get("setup_s")))
PY
done
# the bench line once more, now that this batch's own PMC summary is in profiles/ form (roofline.traffic is read from it)
cp gpurun_out/${T}_pmc_traffic.json gpurun_out/${T}_pmc_traffic.txt profiles/ 2>/dev/null
python bench.py > gpurun_out/${T}_bench_config3.json 2> /dev/null
MFM_RES_PROF=5 python bench.py --steps 10 --warmup 3 --no-other-configs --cpu-seconds 0 --fit-iters 0 2>&1 | grep -A30 "resident profile" | cut -c1-330 > gpurun_out/${T}_res_phases.txt
