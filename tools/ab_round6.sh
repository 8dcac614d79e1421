#!/bin/bash
# usage (GPU box): tools/ab_round6.sh <rounds> [workload args] -- "<label>=<dir>:<ENV=1 ...>" ...
# same-box A/B of bench.py, interleaved: <dir> is `.` (this tree) or `_ab/base` (an archive of the round-5 head d8d2e82 with its own
# library: `git archive d8d2e82 | tar -x -C _ab/base` + `python -m oneshotdet_amd.build` there; git-ignored).  Example:
#   tools/ab_round6.sh 3 -- "base=_ab/base:X=1" "current=.:X=1" "no owner stores=.:OSD_WGRAD_NO_OWNER=1"
R=$1; shift
ARGS="--no-cpu-baseline --no-conv-timing --steps 30 --warmup 8"
while [ "$1" != "--" ] && [ $# -gt 0 ]; do ARGS="$ARGS $1"; shift; done
shift
for i in $(seq $R); do
  for spec in "$@"; do
    label=${spec%%=*}; rest=${spec#*=}; dir=${rest%%:*}; envs=${rest#*:}
    L=$(cd $GRAFT_REPO_ROOT/$dir && env $envs python3 bench.py $ARGS 2>/dev/null | grep '^{"metric"' | tail -1)
    echo "$label [$dir; $envs]: $(echo "$L" | python3 -c 'import sys,json; j=json.loads(sys.stdin.read()); print(j["value"], "img/s", j["ms_per_step"], "ms/step", (j.get("step_ms") or {}))')"
  done
done
