#!/bin/bash
# usage (GPU box): tools/ab_gn1p.sh [rounds] — same-box A/B of the training step with the one-pass GroupNorm kernels (groupnorm_onepass.hip)
# against the two-launch kernels, interleaved: OSD_GN_ONEPASS = 0 | 1 | f (forward only) | b (backward only), and the pixels per thread
R=${1:-3}
ARGS="--no-cpu-baseline --no-conv-timing --steps 30 --warmup 8"
run() {  # label, env...
  local label=$1; shift
  L=$(cd $GRAFT_REPO_ROOT && env "$@" python3 bench.py $ARGS 2>/dev/null | grep '^{"metric"' | tail -1)
  echo "$label: $(echo "$L" | python3 -c 'import sys,json; j=json.loads(sys.stdin.read()); print(j["value"], "img/s", j["ms_per_step"], "ms/step", (j.get("step_ms") or {}).get("median"))')"
}
for i in $(seq $R); do
  run "two-launch GroupNorm" OSD_GN_ONEPASS=0
  run "one-pass forward only, 16 pixels per thread" OSD_GN_ONEPASS=f
  run "one-pass forward only, 18 pixels per thread (496 workgroups: one round)" OSD_GN_ONEPASS=f OSD_GN1P_U_FWD=18
  run "one-pass forward only, 20 pixels per thread (440 workgroups)" OSD_GN_ONEPASS=f OSD_GN1P_U_FWD=20
  run "one-pass forward + backward (8 pixels per thread)" OSD_GN_ONEPASS=1 OSD_GN1P_U_BWD=8
done
