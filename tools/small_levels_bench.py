"""GPU box: one FCOS tower layer over the SMALL FPN levels (P5 + P6 + P7 of 4 x 1280 x 1280 images: 8,400 pixels, K = 2,304, 256 channels;
16 launches per training step) on every conv algorithm, forward and data-gradient form (ReLU mask), back to back and with cold weights.  python tools/small_levels_bench.py [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oneshotdet_amd import ops, _lib

dt = torch.bfloat16
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 4
sizes = [(40, 40), (20, 20), (10, 10)]
xs = [torch.randn(bs, h, w, 256, device="cuda").to(dt) for h, w in sizes]
ms = [(torch.rand(bs, h, w, 256, device="cuda") > 0.5).to(dt) for h, w in sizes]
wt = torch.randn(256, 256, 3, 3, device="cuda") / 48.0
pc = ops.pack_conv(wt, bias=torch.zeros(256, device="cuda"), dtype=dt)
REPS = 20
flush = torch.zeros(150 * 1024 * 1024, device="cuda")
M = sum(bs * h * w for h, w in sizes)
fl = 2.0 * M * 256 * 2304


def name(algo):
    a0 = algo - 1
    return {56: "deep 64x32x8", 57: "deep 64x64x5", 58: "deep 64x64x8", 59: "deep 32x64x8"}.get(a0, "%s v%d t%d" % ("dma" if a0 < 32 else "reg", (a0 >> 3) & 3, a0 & 7))


for label, masks in (("forward", None), ("data gradient (mask)", ms)):
    res = []
    for algo in sorted(set(ops.conv_algo_candidates(256, False, has_mask=masks is not None) + [57, 58, 59, 60])):
        try:
            ops.conv2d_multi(xs, [pc] * 3, stride=1, pad=1, masks=masks, algo=algo, _whole=True)
        except _lib.OsdError:
            continue
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(REPS):
            ops.conv2d_multi(xs, [pc] * 3, stride=1, pad=1, masks=masks, algo=algo, _whole=True)
        b.record()
        torch.cuda.synchronize()
        warm = a.elapsed_time(b) / REPS * 1e3
        cold = 0.0
        for _ in range(8):          # weights cold (600 MB pass in between), activations re-touched: what the launch meets inside a step
            flush.add_(1.0)
            for x in xs:
                x.float().sum()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            ops.conv2d_multi(xs, [pc] * 3, stride=1, pad=1, masks=masks, algo=algo, _whole=True)
            b.record()
            torch.cuda.synchronize()
            cold += a.elapsed_time(b) * 1e3 / 8
        res.append((cold, warm, name(algo), algo))
    res.sort()
    print("%s, M = %d grouped over 3 levels:" % (label, M))
    for c, t, nm, algo in res[:12]:
        print("   %-16s (algo %2d) cold %6.1f us  warm %6.1f us  %5.0f TFLOP/s" % (nm, algo, c, t, fl / t / 1e6))
