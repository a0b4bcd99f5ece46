#!/bin/bash
# exact latent draws on the device: rates next to the Philox default (same session shapes as scripts/r06_exact_baseline.sh)
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
O=gpurun_out/r06_c_exact_device.txt; : > $O
export MFM_LATENT_TIMING=1
run() { echo "### $*" >> $O; python bench.py --gpus 1 --fit-iters 0 --no-other-configs --no-kernel-timing --cpu-seconds 0 "$@" 2>gpurun_out/r06_c_err.log | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(json.dumps({k:d[k] for k in ('value','ms_per_step','steps')}), d['config']['workload'][:100])" >> $O; grep "^\[latent\]" gpurun_out/r06_c_err.log | tail -2 >> $O; grep -v "^\[latent\]" gpurun_out/r06_c_err.log | grep -v amdgpu.ids | tail -3 >> $O; }
run --config 2 --task classification --latent exact --steps 100 --warmup 5
run --config 2 --task ordered --latent exact --steps 50 --warmup 5
run --config 3 --task classification --latent exact --steps 30 --warmup 3
run --config 3 --task ordered --latent exact --steps 30 --warmup 3
run --config 5 --scale 0.1 --latent exact --steps 6 --warmup 2
run --config 5 --scale 1.0 --latent exact --steps 5 --warmup 2
cat $O
