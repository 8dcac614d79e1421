#!/bin/bash
# usage (GPU box): tools/ab_lib.sh <rounds> <label>=<path to another build of the library> ... — same-box A/B of bench.py with the default
# library against tagged builds (OSD_BUILD_TAG=<name> OSD_BUILD_FLAGS=... python -m oneshotdet_amd.build), interleaved.
R=$1; shift
ARGS="--no-cpu-baseline --no-conv-timing --steps 30 --warmup 8"
one() { L=$(cd $GRAFT_REPO_ROOT && env "$@" python3 bench.py $ARGS 2>/dev/null | grep '^{"metric"' | tail -1); echo "$L" | python3 -c 'import sys,json; j=json.loads(sys.stdin.read()); print(j["value"], "img/s", j["ms_per_step"], "ms/step", (j.get("step_ms") or {}).get("median"))'; }
for i in $(seq $R); do
  echo "default: $(one X=1)"
  for spec in "$@"; do echo "${spec%%=*}: $(one OSD_LIB_PATH=$GRAFT_REPO_ROOT/${spec#*=})"; done
done
