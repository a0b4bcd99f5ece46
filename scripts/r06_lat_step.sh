# latent draws after a change of the walker step: the exact-latent tests, then the timing lines at config 5 and at config 3's shape
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_exact_latent.py tests/test_gpu_task_kernels.py -x -q 2>&1 | tail -5
MFM_LATENT_TIMING=1 python bench.py --config 5 --scale 1.0 --steps 3 --warmup 1 --cpu-seconds 0 --fit-iters 0 --long-seconds 0 2>&1 | grep -E "^\[latent\]|\"value\"" | tail -3 | cut -c1-420
MFM_LATENT_TIMING=1 python bench.py --config 3 --task classification --steps 10 --warmup 2 --cpu-seconds 0 --fit-iters 0 --long-seconds 0 --no-other-configs 2>&1 | grep -E "^\[latent\]|\"value\"" | tail -2 | cut -c1-420
