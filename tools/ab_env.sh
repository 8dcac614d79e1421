#!/bin/bash
# usage (GPU box): tools/ab_env.sh <rounds> "<label>=<ENV=1 ENV2=x>" ... — same-box A/B of bench.py under sets of environment switches,
# interleaved; the first entry is usually "default=X=1".
R=$1; shift
ARGS="--no-cpu-baseline --no-conv-timing --steps 30 --warmup 8"
for i in $(seq $R); do
  for spec in "$@"; do
    label=${spec%%=*}; envs=${spec#*=}
    L=$(cd $GRAFT_REPO_ROOT && env $envs python3 bench.py $ARGS 2>/dev/null | grep '^{"metric"' | tail -1)
    echo "$label [$envs]: $(echo "$L" | python3 -c 'import sys,json; j=json.loads(sys.stdin.read()); print(j["value"], "img/s", j["ms_per_step"], "ms/step", (j.get("step_ms") or {}).get("median"))')"
  done
done
