#!/bin/bash
# usage (GPU box): tools/prof_pmc.sh <tag> <bench args...>  -> gpurun_out/pmc_<tag>/{fetch,write}/...
# PMC counters are collected in their OWN runs (no sys/hip/hsa tracing), one pass per counter (TCC slots).
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/$C -o run -- python3 $GRAFT_REPO_ROOT/bench.py "$@" > $OUT/$C.log 2>&1
  tail -1 $OUT/$C.log | cut -c1-200
done
ls -R $OUT | head -30
