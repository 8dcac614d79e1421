"""GPU box: the latency-sized convs of the query backbone (127 x 127 queries: M = 128 .. 2,048 pixels at bs 8) with COLD weights — a
600 MB elementwise pass between two launches evicts L2 and the Infinity Cache, as ~1 GB of parameter state does between two uses
of a weight inside a training step — against the same launch warm (back to back).  Every 64 x 64-tile algorithm; OSD_DMA_DEEP=5 / 8
adds the deep rings.  python tools/small_m_cold.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oneshotdet_amd import ops, _lib

dt = torch.bfloat16
shapes = [(8, 8, 8, 256, 256, 3, 1), (8, 4, 4, 512, 512, 3, 1), (8, 8, 8, 1024, 256, 1, 0), (8, 4, 4, 2048, 512, 1, 0), (8, 16, 16, 128, 128, 3, 1),
          (8, 4, 4, 512, 2048, 1, 0), (8, 8, 8, 256, 1024, 1, 0)]
flush = torch.zeros(150 * 1024 * 1024, device="cuda")
REPS = 12
algos = [1 + v * 8 + 2 for v in (0, 1, 2)] + [57, 58, 59, 60]
for (n, h, w, cin, cout, k, p) in shapes:
    x = torch.randn(n, h, w, cin, device="cuda").to(dt)
    wt = torch.randn(cout, cin, k, k, device="cuda") / (cin * k * k) ** 0.5
    pc = ops.pack_conv(wt, bias=torch.zeros(cout, device="cuda"), dtype=dt)
    res = []
    for algo in algos:
        try:
            y = ops.conv2d(x, pc, stride=1, pad=p, algo=algo)
        except _lib.OsdError:
            continue
        torch.cuda.synchronize()
        cold = 0.0
        for _ in range(REPS):
            flush.add_(1.0)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            ops.conv2d(x, pc, stride=1, pad=p, algo=algo, out=y)
            b.record()
            torch.cuda.synchronize()
            cold += a.elapsed_time(b)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(REPS):
            ops.conv2d(x, pc, stride=1, pad=p, algo=algo, out=y)
        b.record()
        torch.cuda.synchronize()
        a0 = algo - 1
        res.append((cold / REPS * 1e3, a.elapsed_time(b) / REPS * 1e3, ("%s v%d t%d" % ("dma", (a0 >> 3) & 3, a0 & 7)) if a0 < 32 else {56: "64x32x8", 57: "64x64x5", 58: "64x64x8", 59: "32x64x8"}[a0]))
    res.sort()
    print("M=%4d N=%4d K=%4d  cold / warm us: " % (n * h * w, cout, cin * k * k) + "  ".join("%s %.1f/%.1f" % (nm, c, wm) for c, wm, nm in res))
