"""GPU box: the second stage's ROI-pool backward (`osd_roi_pool_levels_bwd`) in isolation, on (a) the sampled boxes a
second-stage bench step really produces, (b) synthetic size mixes.  Prints the FPN-level histogram and us per call.
    python tools/roi_bwd_bench.py"""
import math, os, sys
sys.path.insert(0, os.getcwd())
import torch
from oneshotdet_amd import ops, spec, synth, train

dt = torch.bfloat16
shapes = [(100, 128), (50, 64), (25, 32), (13, 16), (7, 8)]


def time_it(fn, reps=10):
    fn(); torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(reps):
        fn()
    ev[1].record(); torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / reps * 1e3


def levels(b):
    s = torch.sqrt((b[:, 2] - b[:, 0] + 1) * (b[:, 3] - b[:, 1] + 1))
    k = torch.floor(4 + torch.log2(s / 224 + 1e-6)).clamp(3, 7).long() - 3
    return torch.bincount(k, minlength=5).tolist()


def run(name, boxes, counts):
    n, r, _ = boxes.shape
    dy = (torch.randn(n * r, 7, 7, 256, device="cuda") * 0.1).to(dt)
    us = time_it(lambda: ops.roi_pool_levels_bwd(shapes, spec.POOLER_SCALES, boxes, counts, dy, 7, 2))
    valid = torch.cat([boxes[i, :int(counts[i])] for i in range(n)]).cpu()
    wh = torch.stack([valid[:, 2] - valid[:, 0], valid[:, 3] - valid[:, 1]], 1)
    print("%-34s %7.1f us   levels P3..P7 %s   median w x h %.0f x %.0f" % (name, us, levels(valid), wh[:, 0].median(), wh[:, 1].median()))


g = torch.Generator().manual_seed(0)
for name, lo, hi in (("small (16-96 px)", 16, 96), ("medium (96-320 px)", 96, 320), ("large (320-800 px)", 320, 800), ("mixed (16-800 px)", 16, 800)):
    wh = torch.rand(8, 128, 2, generator=g) * (hi - lo) + lo
    xy = torch.rand(8, 128, 2, generator=g) * torch.tensor([1024.0, 800.0])
    b = torch.cat([xy, xy + wh], 2)
    b[..., 0::2] = b[..., 0::2].clamp(0, 1023)
    b[..., 1::2] = b[..., 1::2].clamp(0, 799)
    run(name, b.cuda().contiguous(), torch.full((8,), 128, dtype=torch.int32, device="cuda"))

