#!/bin/bash
# usage: scripts/prof_cfg.sh <tag> <bench args...>  -- rocprofv3 kernel trace + stats of one bench.py run, summary under gpurun_out/
tag=$1; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof_$tag -- python $R/bench.py "$@" --cpu-iters 0 --fit-iters 0 --no-kernel-timing > /tmp/prof_$tag.log 2>&1
cd $R
python scripts/rocpd_summary.py $(ls /tmp/prof_$tag/*/*_results.db | head -1) gpurun_out/prof_$tag.txt "rocprofv3 --kernel-trace --stats -- python bench.py $* --cpu-iters 0 --fit-iters 0 --no-kernel-timing" | cut -c1-200 | head -30
