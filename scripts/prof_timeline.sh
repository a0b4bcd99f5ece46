#!/bin/bash
# usage: scripts/prof_timeline.sh <tag> <marker kernel> <bench args...>
tag=$1; marker=$2; shift; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace -d /tmp/tl_$tag -- python $R/bench.py "$@" --cpu-iters 0 --fit-iters 0 --no-kernel-timing > /tmp/tl_$tag.log 2>&1
cd $R
python scripts/trace_timeline.py $(ls /tmp/tl_$tag/*/*_results.db | head -1) $marker $TL_FLAGS > gpurun_out/timeline_$tag.txt 2>&1
head -60 gpurun_out/timeline_$tag.txt
