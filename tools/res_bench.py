"""GPU box: the HBM-bound 1x1 convs of the bottlenecks (bs 8) with and without their residual operand, per algorithm: how much
of their time is the residual's bytes and how much is the epilogue waiting for them.   python tools/res_bench.py"""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from oneshotdet_amd import ops

g = torch.Generator(device="cuda").manual_seed(0)
rnd = lambda *s: (torch.randn(*s, device="cuda", generator=g) * 0.5)


def t(fn, reps=20):
    fn(); torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(reps):
        fn()
    ev[1].record(); torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / reps * 1e3


for (n, h, w, cin, cout) in ((8, 50, 64, 256, 1024), (8, 100, 128, 128, 512), (8, 25, 32, 512, 2048)):
    x = rnd(n, h, w, cin).bfloat16()
    res = rnd(n, h, w, cout).bfloat16()
    pc = ops.pack_conv(rnd(cout, cin, 1, 1) / cin ** 0.5, bias=torch.zeros(cout, device="cuda"), dtype=torch.bfloat16)
    mb_res = (x.numel() + 2 * res.numel()) * 2 / 1e6
    mb_plain = (x.numel() + res.numel()) * 2 / 1e6
    rows = []
    for algo in ops.conv_algo_candidates(pc.cout_store, False):
        try:
            a = t(lambda: ops.conv2d(x, pc, act=ops.ACT_RELU, res=res, res_mode=ops.RES_SAME, algo=algo))
            b = t(lambda: ops.conv2d(x, pc, act=ops.ACT_RELU, algo=algo))
        except Exception:
            continue
        rows.append((a, b, algo))
    rows.sort()
    print("1x1 %d -> %d, M = %d: %.0f MB with residual, %.0f MB without" % (cin, cout, n * h * w, mb_res, mb_plain))
    for a, b, algo in rows[:5]:
        print("   algo %3d (tile %d variant %d): with residual %.1f us = %.2f TB/s, without %.1f us = %.2f TB/s"
              % (algo, (algo - 1) & 7, ((algo - 1) >> 3) & 3, a, mb_res / a, b, mb_plain / b))
