"""GPU box: every weight-gradient tuner candidate (variant x split-target code) on the REAL stage queues of a bs=8 bf16 training step
(what `BackwardPass._flush_wgrads` launches as one mixed-geometry launch per stage), timed back to back on one stream.
python tools/mixed_wgrad_candidates.py [reps]      (OSD_WGRAD_NO_OWNER=1: the same with atomics only)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from oneshotdet_amd import ops, spec, synth, train, train_backward

REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 6
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
eng.train_step(images, queries, gtb, gtc)
eng.join()
torch.cuda.synchronize()
train_backward.BackwardPass._flush_wgrads = orig
TARGETS = [512, 256, 128, 64, 1024, 768, 1536, 2048]
for j, which, q in captured:
    if j != 0:
        continue                     # the target backbone's stages
    part = q[:24]
    items = [(x, dy, torch.empty_like(c.gw), c.bn_scale, torch.empty_like(c.gb) if c.has_bias else None, c.r, c.s, stride, pad, c.cout)
             for c, x, dy, stride, pad in part]
    gf = sum(2.0 * dy.shape[0] * dy.shape[1] * dy.shape[2] * c.cout * c.r * c.s * x.shape[-1] for c, x, dy, _, _ in part) / 1e9
    print("== %d convs, %.1f GFLOP: %s" % (len(part), gf, ", ".join(sorted(set("%s" % c.name.split(".")[1] for c, *_ in part)))), flush=True)
    cands = ops.wgrad_algo_candidates(ops.OSD_BF16, max(c.cout for c, *_ in part), max(x.shape[-1] for _, x, *_ in part))
    res = []
    for cand in cands:
        v, t = (cand - 1) & 15, (cand - 1) >> 4
        try:
            ops.conv2d_wgrad_mixed(items, algo=cand)
        except Exception:      # noqa: BLE001
            continue
        torch.cuda.synchronize()
        best = float("inf")
        for _ in range(2):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(REPS):
                ops.conv2d_wgrad_mixed(items, algo=cand)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / REPS * 1e3)
        res.append((best, v, t))
    res.sort()
    for us, v, t in res[:14]:
        print("   variant %2d target %4d: %7.1f us  %5.0f TFLOP/s" % (v, TARGETS[t], us, gf / us * 1e3))
    by_v = {}
    for us, v, t in res:
        by_v.setdefault(v, []).append((TARGETS[t], round(us, 1)))
    for v in sorted(by_v):
        print("   variant %2d: %s" % (v, sorted(by_v[v])))
