#!/bin/bash
# usage (GPU box): tools/ab_streams.sh [rounds] — same-box A/B of which engine roles share a stream / a hardware queue.
R=${1:-2}
ARGS="--no-cpu-baseline --no-conv-timing --steps 30 --warmup 8"
run() {  # label, env...
  local label=$1; shift 1
  L=$(cd $GRAFT_REPO_ROOT && env "$@" python3 bench.py $ARGS 2>/dev/null | grep '^{"metric"' | tail -1)
  echo "$label: $(echo "$L" | python3 -c 'import sys,json; j=json.loads(sys.stdin.read()); print(j["value"], "img/s", j["ms_per_step"], "ms/step", (j.get("step_ms") or {}).get("median"))')"
}
for i in $(seq $R); do
  run "default (6 side streams, order main,s1,u,w,w2,p)" X=1
  run "update on weight-gradient stream 1 (u=w)" OSD_STREAM_ALIAS=u=w
  run "update on weight-gradient stream 2 (u=w2)" OSD_STREAM_ALIAS=u=w2
  run "proposals on weight-gradient stream 1 (p=w)" OSD_STREAM_ALIAS=p=w
  run "proposals on weight-gradient stream 2 (p=w2)" OSD_STREAM_ALIAS=p=w2
  run "update and proposals on the weight-gradient streams (u=w,p=w2)" OSD_STREAM_ALIAS=u=w,p=w2
  run "update and proposals on the weight-gradient streams (u=w2,p=w)" OSD_STREAM_ALIAS=u=w2,p=w
  run "one weight-gradient stream (w2=w)" OSD_STREAM_ALIAS=w2=w
  run "one weight-gradient stream, update on it (w2=w,u=w)" OSD_STREAM_ALIAS=w2=w,u=w
done
