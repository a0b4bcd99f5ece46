#!/bin/bash
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export MFM_LATENT_TIMING=1
timeout 1200 python -m pytest tests/test_gpu_exact_latent.py -x -q 2>&1 | grep -v "^\[latent\]" | tail -25
echo "=== host rng parity (mode exact via exact_latent_draws)"; timeout 600 python -m pytest tests/test_gpu_host_rng_parity.py -x -q 2>&1 | grep -v "^\[latent\]" | tail -5
