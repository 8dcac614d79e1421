#!/bin/bash
# usage (GPU box): tools/ab_round4.sh [rounds] — same-box A/B of this tree against _ab/base (an archive of an earlier commit with its
# own library) and against itself with single features switched off, interleaved.
R=${1:-2}
ARGS="--no-cpu-baseline --no-conv-timing --steps 30 --warmup 8"
run() {  # label, dir, env...
  local label=$1 dir=$2; shift 2
  L=$(cd $GRAFT_REPO_ROOT/$dir && env "$@" python3 bench.py $ARGS 2>/dev/null | tail -1)
  echo "$label: $(echo "$L" | python3 -c 'import sys,json; j=json.loads(sys.stdin.read()); print(j["value"], "img/s", j["ms_per_step"], "ms/step", (j.get("step_ms") or {}).get("median"))')"
}
for i in $(seq $R); do
  run "base(r3 head)" _ab/base X=1
  run "current" . X=1
  run "current, epilogue not pipelined" . OSD_LIB_PATH=$GRAFT_REPO_ROOT/oneshotdet_amd/lib/liboneshotdet_hip_nopipe.so
  run "current, towers' gradient sum as add_mask launches" . OSD_NO_HEAD_SUM_FUSION=1
  run "current, SGD and forward repack as separate launches (and a gradient memset per step)" . OSD_NO_FUSED_REPACK=1
  run "current, gradient memset per step" . OSD_NO_CONSUME_GRADS=1
  run "current, proposals on the bbox tower's hardware queue (the natural first-use order)" . OSD_WARM_ORDER=main,w,s1,w2,p,u
done
