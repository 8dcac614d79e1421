#!/bin/bash
# usage (GPU box): tools/tuner_variance.sh <runs> "<bench.py arguments>" — every run tunes afresh and saves its picks; afterwards the fastest and
# the slowest run's picks are replayed twice each (no tuning): if the replays keep the ranking, the picks are what differs
N=${1:-5}; EXTRA=$2
O=$GRAFT_REPO_ROOT/gpurun_out/tv; rm -rf $O; mkdir -p $O
ARGS="--no-cpu-baseline --no-conv-timing --steps 30 --warmup 9 $EXTRA"
cd $GRAFT_REPO_ROOT
for i in $(seq $N); do
  L=$(OSD_TUNER_CACHE=$O/picks_$i.json python3 bench.py $ARGS 2>/dev/null | grep '^{"metric"' | tail -1)
  echo "$i $(echo "$L" | python3 -c 'import sys,json; j=json.loads(sys.stdin.read()); print(j["value"], j["ms_per_step"])')" | tee -a $O/runs.txt
done
BEST=$(sort -k2 -n -r $O/runs.txt | head -1 | cut -d" " -f1); WORST=$(sort -k2 -n $O/runs.txt | head -1 | cut -d" " -f1)
echo "fastest run $BEST, slowest run $WORST"
for r in 1 2; do
  for w in $BEST $WORST; do
    L=$(OSD_TUNER_CACHE=$O/picks_$w.json python3 bench.py $ARGS 2>/dev/null | grep '^{"metric"' | tail -1)
    echo "replay of run $w: $(echo "$L" | python3 -c 'import sys,json; j=json.loads(sys.stdin.read()); print(j["value"], j["ms_per_step"])')"
  done
done
python3 - <<PY
import json
a=json.load(open("$O/picks_$BEST.json")); b=json.load(open("$O/picks_$WORST.json"))
for name in a:
    da, db = dict(map(tuple, a[name])), dict(map(tuple, b[name]))
    for k in da:
        if k in db and da[k] != db[k]:
            print(name, "fast", da[k], "slow", db[k], k[:150])
PY
