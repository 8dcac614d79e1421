#!/bin/bash
# usage (GPU box): tools/ab_bench.sh [rounds] [bench args...] — same-box A/B of two trees of this repo: _ab/base (a `git archive` of
# an earlier commit with its own built library) against the working tree, interleaved, so that the box-to-box spread of the
# pool (+-5 %) does not enter the comparison.  Prints images/s and ms/step per run.
R=${1:-2}; shift
ARGS=${@:---no-cpu-baseline --no-conv-timing --steps 30 --warmup 8}
for i in $(seq $R); do
  for T in _ab/base .; do
    L=$(cd $GRAFT_REPO_ROOT/$T && python3 bench.py $ARGS 2>/dev/null | tail -1)
    echo "$T: $(echo "$L" | python3 -c 'import sys,json; j=json.loads(sys.stdin.read()); print(j["value"], "img/s", j["ms_per_step"], "ms/step")')"
  done
done
