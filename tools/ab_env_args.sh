#!/bin/bash
# usage (GPU box): tools/ab_env_args.sh <rounds> "<bench.py arguments>" "<label>=<ENV=1 ENV2=x>" ... — tools/ab_env.sh with the workload chosen by
# the caller (e.g. "--workload config5", "--mode forward --dtype bf16", "--second-stage")
R=$1; shift
EXTRA=$1; shift
ARGS="--no-cpu-baseline --no-conv-timing --steps 30 --warmup 8 $EXTRA"
for i in $(seq $R); do
  for spec in "$@"; do
    label=${spec%%=*}; envs=${spec#*=}
    L=$(cd $GRAFT_REPO_ROOT && env $envs python3 bench.py $ARGS 2>/dev/null | grep '^{"metric"' | tail -1)
    echo "$label [$envs]: $(echo "$L" | python3 -c 'import sys,json; j=json.loads(sys.stdin.read()); print(j["value"], "img/s", j["ms_per_step"], "ms/step", (j.get("step_ms") or {}).get("median"))')"
  done
done
