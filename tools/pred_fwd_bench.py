"""GPU box: the FCOS prediction convs' forward (3x3 256 -> 4 over P3..P7 at bs 8) — the patch kernel (conv_pred.hip, algo 51) against
the LDS-DMA tile kernels, one grouped launch over the five levels, and per level.  python tools/pred_fwd_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oneshotdet_amd import ops, _lib

sizes = [(8, 100, 128), (8, 50, 64), (8, 25, 32), (8, 13, 16), (8, 7, 8)]
xs = [torch.relu(torch.randn(n, h, w, 256, device="cuda")).bfloat16() for n, h, w in sizes]
pc = ops.pack_conv(torch.randn(4, 256, 3, 3, device="cuda") / 48, bias=torch.zeros(4, device="cuda"), dtype=torch.bfloat16)


def t(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


for algo in [51] + [1 + v * 8 + tl for v in (0, 1, 2, 3) for tl in (3, 2)]:
    try:
        us = t(lambda: ops.conv2d_grouped(xs, pc, pad=1, algo=algo, _whole=True))
    except _lib.OsdError as e:
        print("algo %d refused: %s" % (algo, str(e)[:80]))
        continue
    per = []
    for x in xs[:2]:
        try:
            per.append("%.1f" % t(lambda: ops.conv2d(x, pc, pad=1, algo=algo)))
        except _lib.OsdError:
            per.append("-")
    print("algo %2d: all five levels %.1f us; P3 / P4 alone %s us" % (algo, us, " / ".join(per)))
