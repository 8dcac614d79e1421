#!/bin/bash
# usage (GPU box): tools/prof_bench.sh <tag> <bench args...>   -> gpurun_out/prof_<tag>/
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o run -- python3 $GRAFT_REPO_ROOT/bench.py "$@" > $OUT/bench.log 2>&1
tail -1 $OUT/bench.log
ls -R $OUT | head -20
