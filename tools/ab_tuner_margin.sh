#!/bin/bash
# usage (GPU box): tools/ab_tuner_margin.sh [rounds] — the tuner's photo finish (candidates within 5 % re-timed twice) against single-bracket
# picks (OSD_TUNE_MARGIN=0): every run tunes afresh, so the spread over runs is the cost of wrong picks
R=${1:-5}
ARGS="--no-cpu-baseline --no-conv-timing --steps 30 --warmup 8"
run() {
  local label=$1; shift
  T0=$(date +%s.%N)
  L=$(cd $GRAFT_REPO_ROOT && env "$@" python3 bench.py $ARGS 2>/dev/null | grep '^{"metric"' | tail -1)
  T1=$(date +%s.%N)
  echo "$label: $(echo "$L" | python3 -c 'import sys,json; j=json.loads(sys.stdin.read()); print(j["value"], "img/s", j["ms_per_step"], "ms/step")') wall $(python3 -c "print(round($T1-$T0,1))") s"
}
for i in $(seq $R); do
  run "single bracket" OSD_TUNE_MARGIN=0
  run "photo finish" X=1
done
