"""GPU box: how should a ResNet stage's weight gradients be grouped into launches?  Captures the per-stage queues of one real
bs=8 bf16 training step (what `BackwardPass._flush_wgrads` launches as ONE mixed-geometry launch) and times, isolated and
back to back on one stream, each tuned:
  mixed     the single mixed launch (today's default);
  by-shape  one launch per conv shape (`conv2d_wgrad_multi`: same channels / kernel / stride, so the software-pipelined kernel's
            team mode applies; single convs through `conv2d_wgrad`);
  wide/rest the 256-wide convs (cin >= 256 and cout >= 256) in one mixed launch, the rest in another;
  3x3/1x1   one mixed launch per kernel size.
python tools/stage_wgrad_groups.py [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from oneshotdet_amd import ops, spec, synth, train, train_backward

REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 20
B = 8
captured = []
orig = train_backward.BackwardPass._flush_wgrads


def spy(self, j, which):
    if self._wqs[j]:
        captured.append((j, which, list(self._wqs[j])))
    return orig(self, j, which)


train_backward.BackwardPass._flush_wgrads = spy
eng = train.TrainEngine(synth.make_state_dict(spec.hot_path_shapes()), dtype=torch.bfloat16)
images = torch.from_numpy(synth.make_images("bench.target", B, 800, 1024, seed=1000)).cuda()
queries = torch.from_numpy(synth.make_images("bench.query", B, 127, 127, seed=1000)).cuda()
gts = synth.make_gt_boxes(B, 800, 1024, seed=1000, max_boxes=6)
gtb = np.zeros((B, max(len(g) for g in gts), 4), dtype=np.float32)
for i, g in enumerate(gts):
    gtb[i, :len(g)] = g
gtb, gtc = torch.from_numpy(gtb).cuda(), torch.tensor([len(g) for g in gts], dtype=torch.int32).cuda()
with ops.tuning():
    eng.train_step(images, queries, gtb, gtc)
eng.join()
torch.cuda.synchronize()
train_backward.BackwardPass._flush_wgrads = orig


def mixed_items(part):
    return [(x, dy, c.gw, c.bn_scale, c.gb if c.has_bias else None, c.r, c.s, stride, pad, c.cout) for c, x, dy, stride, pad in part]


def launch_group(part):
    """One launch for convs of one shape (or a single conv)."""
    c0, x0, dy0, stride, pad = part[0]
    if len(part) == 1:
        return lambda: ops.conv2d_wgrad(x0, dy0, c0.gw, c0.r, c0.s, stride, pad, c0.cout, scale=c0.bn_scale, db=c0.gb if c0.has_bias else None)
    items = [(x, dy, c.gw, c.bn_scale, c.gb if c.has_bias else None) for c, x, dy, _, _ in part]
    return lambda: ops.conv2d_wgrad_multi(items, c0.r, c0.s, stride, pad, c0.cout)


def launch_mixed(part):
    if len(part) == 1:
        return launch_group(part)
    items = mixed_items(part)
    return lambda: ops.conv2d_wgrad_mixed(items)


def timed(fns):
    with ops.tuning():
        for f in fns:
            f()
    torch.cuda.synchronize()
    for f in fns:
        f()
    torch.cuda.synchronize()
    best = float("inf")
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(REPS):
            for f in fns:
                f()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / REPS * 1e3)
    return best


def flops(part):
    return sum(2.0 * dy.shape[0] * dy.shape[1] * dy.shape[2] * c.cout * c.r * c.s * x.shape[-1] for c, x, dy, _, _ in part)


def partition(part, keyf):
    groups = {}
    for it in part:
        groups.setdefault(keyf(it), []).append(it)
    return list(groups.values())


def shape_key(it):
    c, x, dy, stride, pad = it
    return (x.shape[-1], c.cout, c.r, c.s, stride, pad)


for j, which, q in captured:
    for i in range(0, len(q), 24):
        part = q[i:i + 24]
        gf = flops(part) / 1e9
        print("== backbone %d stream %d: %d convs, %d pixels, %.1f GFLOP" % (j, which, len(part), sum(dy.shape[0] * dy.shape[1] * dy.shape[2] for _, _, dy, _, _ in part), gf), flush=True)
        for c, x, dy, stride, pad in part:
            print("     %4d -> %4d  %dx%d s%d  map %dx%d" % (x.shape[-1], c.cout, c.r, c.s, stride, dy.shape[1], dy.shape[2]))
        plans = {
            "mixed": [launch_mixed(part)],
            "by-shape": [launch_group(g) for g in partition(part, shape_key)],
            "wide/rest": [launch_mixed(g) for g in partition(part, lambda it: it[1].shape[-1] >= 256 and it[0].cout >= 256)],
            "3x3/1x1": [launch_mixed(g) for g in partition(part, lambda it: it[0].r)],
            "by-channels": [launch_mixed(g) for g in partition(part, lambda it: (min(it[1].shape[-1], 256), min(it[0].cout, 256)))],
            "backbone/fpn": [launch_mixed(g) for g in partition(part, lambda it: "fpn" in it[0].name)],
            "backbone/fpn3x3/fpn1x1": [(launch_group(g) if ("fpn" in g[0][0].name and g[0][0].r == 3 and all(x[0].cout == g[0][0].cout and x[3] == g[0][3] for x in g)) else launch_mixed(g))
                                       for g in partition(part, lambda it: ("fpn" in it[0].name, it[0].r if "fpn" in it[0].name else 0, it[3] if "fpn" in it[0].name else 0))],
        }
        for name, fns in plans.items():
            try:
                us = timed(fns)
                print("   %-12s %2d launches  %8.1f us  %6.0f TFLOP/s" % (name, len(fns), us, gf / us * 1e3), flush=True)
            except Exception as e:      # noqa: BLE001
                print("   %-12s failed: %s" % (name, str(e)[:200]), flush=True)
