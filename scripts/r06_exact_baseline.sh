#!/bin/bash
# round 6, before the device-side exact latent draws: what the HOST exact mode costs
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && export MYFM_AMD_EXACT_ON_HOST=1
O=gpurun_out/r06_a_exact_baseline.txt; : > $O
run() { echo "### $*" >> $O; python bench.py --gpus 1 --fit-iters 0 --no-other-configs --no-kernel-timing --cpu-seconds 0 "$@" 2>>gpurun_out/r06_a_err.log | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(json.dumps({k:d[k] for k in ('value','ms_per_step','steps')}), d['config']['workload'][:100])" >> $O; }
run --config 3 --task classification --steps 100 --warmup 5
run --config 3 --task classification --latent host --steps 5 --warmup 1
run --config 3 --task ordered --steps 50 --warmup 5
run --config 3 --task ordered --latent host --steps 5 --warmup 1
run --config 2 --task classification --steps 200 --warmup 10
run --config 2 --task classification --latent host --steps 100 --warmup 5
run --config 2 --task ordered --steps 100 --warmup 10
run --config 2 --task ordered --latent host --steps 50 --warmup 5
run --config 5 --scale 0.1 --steps 6 --warmup 2
run --config 5 --scale 0.1 --latent host --steps 4 --warmup 1
run --config 5 --scale 1.0 --steps 5 --warmup 2
run --config 5 --scale 1.0 --latent host --steps 3 --warmup 1
cat $O
