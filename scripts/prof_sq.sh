#!/bin/bash
# SQ / LDS / TA counters of the kernels of a short bench.py run (rocprofv3 --pmc, counters only: no trace domains)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
tag=${1:-sq}
cd /tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${tag}_a -o p -- python $GRAFT_REPO_ROOT/bench.py ${SQ_BENCH_ARGS:---steps 2 --warmup 1} --cpu-iters 0 --fit-iters 0 --no-kernel-timing > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${tag}_b -o p -- python $GRAFT_REPO_ROOT/bench.py ${SQ_BENCH_ARGS:---steps 2 --warmup 1} --cpu-iters 0 --fit-iters 0 --no-kernel-timing > /dev/null 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_WAIT_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${tag}_c -o p -- python $GRAFT_REPO_ROOT/bench.py ${SQ_BENCH_ARGS:---steps 2 --warmup 1} --cpu-iters 0 --fit-iters 0 --no-kernel-timing > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
for x in a b c; do python scripts/pmc_sq.py gpurun_out/${tag}_$x 2>&1 | tail -14; done
