"""GPU box: marginal in-step cost of kernel families — the bs=8 bf16 training step with the launches of one family skipped at
the C-ABI (timing only; results are garbage).  Interleaved rounds, medians.   python tools/ablate2.py [steps] [rounds]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from oneshotdet_amd import _lib, ops, spec, synth, train

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
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
    eng.train_step(images, queries, gt_boxes, gt_count)
torch.cuda.synchronize()
real = _lib.call
qstream = eng.s1.cuda_stream if eng.s1 is not None else -1
main = torch.cuda.current_stream().cuda_stream


def skipper(pred):
    def call(name, *a):
        if pred(name, torch.cuda.current_stream().cuda_stream, a):
            return None
        return real(name, *a)
    return call


def small_multi(name, st, a):      # grouped / multi forward launches over the small FPN levels only (first problem below 64 x 64 pixels)
    if name not in ("osd_conv2d_fwd_multi", "osd_conv2d_fwd_grouped"):
        return False
    try:
        hs = a[3] if name == "osd_conv2d_fwd_multi" else a[2]
        return False
    except Exception:
        return False


FAM = {
    "nothing skipped": lambda n, s, a: False,
    "GroupNorm backward": lambda n, s, a: n == "osd_groupnorm_relu_bwd_levels",
    "GroupNorm forward": lambda n, s, a: n == "osd_groupnorm_relu_fwd_levels",
    "tower weight gradients (multi)": lambda n, s, a: n == "osd_conv2d_wgrad_multi",
    "backbone/FPN weight gradients (mixed)": lambda n, s, a: n == "osd_conv2d_wgrad_mixed",
    "prediction-conv weight gradient": lambda n, s, a: n == "osd_conv2d_wgrad_pred",
    "everything on the query stream": lambda n, s, a: s == qstream,
    "single-conv launches on the query stream (query backbone)": lambda n, s, a: s == qstream and n in ("osd_conv2d_fwd", "osd_maxpool3x3s2_fwd", "osd_pack_image"),
    "multi / grouped conv launches on the query stream (bbox tower)": lambda n, s, a: s == qstream and n in ("osd_conv2d_fwd_multi", "osd_conv2d_fwd_grouped"),
    "correlation fwd + bwd": lambda n, s, a: n.startswith("osd_correlate"),
    "loss": lambda n, s, a: n.startswith("osd_fcos_loss"),
    "proposals": lambda n, s, a: n.startswith(("osd_proposals", "osd_fcos_score", "osd_append_gt")),
    "update (sgd + repack)": lambda n, s, a: n in ("osd_sgd_momentum_multi", "osd_pack_multi"),
    "scatter / add_mask / upsample bwd": lambda n, s, a: n in ("osd_scatter2x", "osd_add_mask", "osd_upsample2x_bwd"),
}


def timed():
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.train_step(images, queries, gt_boxes, gt_count)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


for _ in range(3):
    eng.train_step(images, queries, gt_boxes, gt_count)
print("unwrapped C-ABI calls: %.3f ms/step" % timed(), flush=True)
res = {k: [] for k in FAM}
for r in range(rounds + 1):
    for k, pred in FAM.items():
        _lib.call = ops._lib.call = skipper(pred)
        for _ in range(2):
            eng.train_step(images, queries, gt_boxes, gt_count)
        t = timed()
        if r > 0:
            res[k].append(t)
_lib.call = ops._lib.call = real
base = float(np.median(res["nothing skipped"]))
for k, v in res.items():
    m = float(np.median(v))
    print("%-42s %7.3f ms/step   saves %6.3f ms   (%s)" % (k, m, base - m, " ".join("%.2f" % x for x in v)))
