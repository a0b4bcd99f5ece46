# round-5 baseline of the conflict-batched chain of config 5 at full size (round-4 kernel): phase stamps + SQ counters
set -x
cd $GRAFT_REPO_ROOT
T=${1:-r05_a}
C5="--config 5 --scale 1.0 --steps 2 --warmup 1 --cpu-iters 0 --fit-iters 0 --no-kernel-timing --long-seconds 0"
MFM_CB_PROF=128 python bench.py $C5 > gpurun_out/${T}_cb_phases.json 2> gpurun_out/${T}_cb_phases.txt
grep k_cb_persist gpurun_out/${T}_cb_phases.txt | tail -3
SQ_BENCH_ARGS="--config 5 --scale 1.0 --steps 1 --warmup 1 --long-seconds 0" bash scripts/prof_sq.sh ${T}_sq5 > gpurun_out/${T}_sq_counters_config5.txt 2>&1
tail -50 gpurun_out/${T}_sq_counters_config5.txt
rm -rf gpurun_out/${T}_sq5_a gpurun_out/${T}_sq5_b gpurun_out/${T}_sq5_c
