"""GPU box: step time of the bs=8 bf16 training step with parts of the work removed (timing only: results are garbage).
Tells how much wall time each kernel family really costs inside the overlapped 4-stream step.

  python tools/ablate.py [steps]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from oneshotdet_amd import ops, spec, synth, train

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
B = 8
eng = train.TrainEngine(synth.make_state_dict(spec.hot_path_shapes()), dtype=torch.bfloat16)
images = torch.from_numpy(synth.make_images("bench.target", B, 800, 1024, seed=1000)).cuda()
queries = torch.from_numpy(synth.make_images("bench.query", B, 127, 127, seed=1000)).cuda()
gts = synth.make_gt_boxes(B, 800, 1024, seed=1000, max_boxes=6)
gtb = np.zeros((B, max(len(g) for g in gts), 4), dtype=np.float32)
for i, g in enumerate(gts):
    gtb[i, :len(g)] = g
gt_boxes = torch.from_numpy(gtb).cuda()
gt_count = torch.tensor([len(g) for g in gts], dtype=torch.int32).cuda()
with ops.tuning():
    eng.forward_backward(images, queries, gt_boxes, gt_count)
torch.cuda.synchronize()


def run(label, **kw):
    for _ in range(3):
        eng.forward_backward(images, queries, gt_boxes, gt_count, **kw)
        eng.optimizer_step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.forward_backward(images, queries, gt_boxes, gt_count, **kw)
        eng.optimizer_step()
    torch.cuda.synchronize()
    print("%-40s %.2f ms/step" % (label, (time.perf_counter() - t0) / steps * 1e3), flush=True)


def run_train_step(label):
    for _ in range(3):
        eng.train_step(images, queries, gt_boxes, gt_count)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.train_step(images, queries, gt_boxes, gt_count)
    torch.cuda.synchronize()
    print("%-40s %.2f ms/step" % (label, (time.perf_counter() - t0) / steps * 1e3), flush=True)


if len(sys.argv) > 2 and sys.argv[2] == "props":
    # interleaved A/B of the training proposals (single runs differ by +-2 %): medians of 6 alternating rounds
    def timed(**kw):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            eng.forward_backward(images, queries, gt_boxes, gt_count, **kw)
            eng.optimizer_step()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e3
    timed(); timed(with_proposals=False)
    a, b = [], []
    for _ in range(6):
        a.append(timed())
        b.append(timed(with_proposals=False))
    a.sort(); b.sort()
    ma, mb = (a[2] + a[3]) / 2, (b[2] + b[3]) / 2
    print("with proposals    median %.3f ms/step  (%s)" % (ma, " ".join("%.2f" % v for v in a)))
    print("without proposals median %.3f ms/step  (%s)" % (mb, " ".join("%.2f" % v for v in b)))
    print("training proposals cost %.2f %% of the step" % (100.0 * (ma - mb) / ma))
    sys.exit(0)
run_train_step("train_step (updates behind backward)")
run("full step")
run_train_step("train_step (updates behind backward)")
run("full step")
run("no training proposals", with_proposals=False)
orig = (ops.conv2d_wgrad, ops.conv2d_wgrad_grouped, ops.conv2d_wgrad_batched, ops.conv2d_wgrad_multi, ops.conv2d_wgrad_mixed)
ops.conv2d_wgrad = lambda *a, **k: None
ops.conv2d_wgrad_grouped = lambda *a, **k: None
ops.conv2d_wgrad_batched = lambda *a, **k: None
ops.conv2d_wgrad_multi = lambda *a, **k: None
ops.conv2d_wgrad_mixed = lambda *a, **k: None
run("no weight gradients")
run("no weight gradients, no proposals", with_proposals=False)
ops.conv2d_wgrad, ops.conv2d_wgrad_grouped, ops.conv2d_wgrad_batched, ops.conv2d_wgrad_multi, ops.conv2d_wgrad_mixed = orig
g0 = (ops.groupnorm_relu_levels, ops.groupnorm_relu_bwd_levels)
saved = eng.wstream, eng.wstream2, eng.s1
eng.wstream = eng.wstream2 = eng.s1 = None
run("single stream")
eng.wstream, eng.wstream2, eng.s1 = saved
# host only: every C-ABI call becomes a no-op (torch allocations and stream ops remain)
real_call = ops._lib.call
ops._lib.call = lambda name, *a: None
run("host only (no kernel launches)")
ops._lib.call = real_call
