#!/bin/bash
# ONE derived counter such as FETCH_SIZE per pass (two: "exceeds the capabilities of the hardware", and rocprofv3 then hangs -> timeout)
# usage (GPU box): tools/prof_pmc_generic.sh <tag> "<counters>" <script.py> [args]  -> gpurun_out/pmcg_<tag>/
TAG=$1; CTRS=$2; shift; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmcg_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d $OUT -o run -- python3 $GRAFT_REPO_ROOT/"$@" > $OUT/run.log 2>&1
tail -2 $OUT/run.log | cut -c1-200
ls $OUT
