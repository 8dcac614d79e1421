#!/bin/bash
# usage (GPU box): tools/tuner_stability.sh [runs] — tune the training step in N fresh processes; per run the step time and the tuner's
# choices; then how many shapes got different choices across the runs (and the cuts of the grouped tower launches).
N=${1:-4}
O=$GRAFT_REPO_ROOT/gpurun_out
for i in $(seq $N); do
  L=$(cd $GRAFT_REPO_ROOT && OSD_DUMP_ALGOS=$O/tuner_run_$i.txt python3 bench.py --no-cpu-baseline --no-conv-timing --steps 30 --warmup 8 2>/dev/null | grep '^{"metric"' | tail -1)
  echo "run $i: $(echo "$L" | python3 -c 'import sys,json; j=json.loads(sys.stdin.read()); print(j["value"], "img/s", j["ms_per_step"], "ms/step")')"
done
python3 - <<PY
import collections
runs = []
for i in range(1, $N + 1):
    d = {}
    for l in open("$O/tuner_run_%d.txt" % i):
        k, _, v = l.rpartition(" -> ")
        d[k] = v.strip()
    runs.append(d)
keys = sorted(set().union(*runs))
diff = [k for k in keys if len(set(r.get(k) for r in runs)) > 1]
print("%d shapes, %d with different choices across the %d runs" % (len(keys), len(diff), len(runs)))
for k in diff:
    print("  %s : %s" % (k[:150], [r.get(k) for r in runs]))
PY
