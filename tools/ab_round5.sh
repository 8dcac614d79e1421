#!/bin/bash
# usage (GPU box): tools/ab_round5.sh [rounds] — same-box A/B of this tree against _ab/base (an archive of the round-4 head, commit
# 192e9a2, with its own library: `git archive 192e9a2 | tar -x -C _ab/base` + `python -m oneshotdet_amd.build` there; git-ignored)
# and against itself with the round's new kernels switched off one at a time, interleaved.
R=${1:-2}
ARGS="--no-cpu-baseline --no-conv-timing --steps 30 --warmup 8"
run() {  # label, dir, env...
  local label=$1 dir=$2; shift 2
  L=$(cd $GRAFT_REPO_ROOT/$dir && env "$@" python3 bench.py $ARGS 2>/dev/null | grep '^{"metric"' | tail -1)
  echo "$label: $(echo "$L" | python3 -c 'import sys,json; j=json.loads(sys.stdin.read()); print(j["value"], "img/s", j["ms_per_step"], "ms/step", (j.get("step_ms") or {}).get("median"))')"
}
for i in $(seq $R); do
  run "base (round-4 head)" _ab/base X=1
  run "current" . X=1
  run "current + conv_px among the tuner candidates" . OSD_PX=1
  run "current, no 128-channel conv_sp tile" . OSD_NO_SP_NARROW=1
done
# config5 (BASELINE.json configs[4]'s per-GPU workload) and forward mode against the base, same box: tools/ab_round5.sh <rounds> config5 | forward
if [ "$2" = "config5" ] || [ "$2" = "forward" ]; then
  [ "$2" = "config5" ] && ARGS="--no-cpu-baseline --no-conv-timing --workload config5 --steps 30 --warmup 9" || ARGS="--no-cpu-baseline --no-conv-timing --mode forward --dtype bf16"
  for i in $(seq $R); do
    run "base (round-4 head), $2" _ab/base X=1
    run "current, $2" . X=1
  done
fi
