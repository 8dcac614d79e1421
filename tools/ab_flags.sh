#!/bin/bash
# usage (GPU box): tools/ab_flags.sh [rounds] ["CFG1" "CFG2" ...] — same-box A/B of the engine's scheduling switches
# (environment variables read by TrainEngine / the library), interleaved with _ab/base (an archived earlier commit).
R=${1:-2}; shift
ARGS="--no-cpu-baseline --no-conv-timing --steps 30 --warmup 8"
if [ $# -eq 0 ]; then set -- "OSD_LOCKSTEP=0" "OSD_LOCKSTEP=1"; fi
for i in $(seq $R); do
  for CFG in "$@"; do
    L=$(cd $GRAFT_REPO_ROOT && env $CFG python3 bench.py $ARGS 2>/dev/null | tail -1)
    echo "$CFG: $(echo "$L" | python3 -c 'import sys,json; j=json.loads(sys.stdin.read()); print(j["value"], "img/s", j["ms_per_step"], "ms/step")')"
  done
  L=$(cd $GRAFT_REPO_ROOT/_ab/base && python3 bench.py $ARGS 2>/dev/null | tail -1)
  echo "base: $(echo "$L" | python3 -c 'import sys,json; j=json.loads(sys.stdin.read()); print(j["value"], "img/s", j["ms_per_step"], "ms/step")')"
done
