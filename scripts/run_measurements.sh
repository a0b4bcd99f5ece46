set -x
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python bench.py --steps 30 --warmup 5 > gpurun_out/r01_i_bench.json 2> gpurun_out/r01_i_bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r01_i_trace -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --cpu-iters 0 > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r01_i_pmc_fetch -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --cpu-iters 0 --no-kernel-timing > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r01_i_pmc_write -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --cpu-iters 0 --no-kernel-timing > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
ls gpurun_out/r01_i_pmc_fetch gpurun_out/r01_i_pmc_write
du -sh gpurun_out/r01_i_pmc_fetch gpurun_out/r01_i_pmc_write
