"""GPU box: the HBM-bound bottleneck 1x1 convs with residual + ReLU (conv3 of every block), best algorithm per shape."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oneshotdet_amd import ops, _lib
for (n, h, w, cin, cout) in [(8, 100, 128, 128, 512), (8, 50, 64, 256, 1024), (8, 200, 256, 64, 256), (8, 25, 32, 512, 2048)]:
    x = torch.randn(n, h, w, cin, device="cuda").bfloat16()
    wt = torch.randn(cout, cin, 1, 1, device="cuda") / cin ** 0.5
    pc = ops.pack_conv(wt, bias=torch.zeros(cout, device="cuda"), dtype=torch.bfloat16)
    res = torch.randn(n, h, w, cout, device="cuda").bfloat16()
    y = torch.empty_like(res)
    byts = (x.numel() + 2 * res.numel()) * 2
    out = []
    for algo in ops.conv_algo_candidates(cout, False):
        try:
            ops.conv2d(x, pc, act=ops.ACT_RELU, res=res, res_mode=ops.RES_SAME, algo=algo, out=y)
        except _lib.OsdError:
            continue
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            ops.conv2d(x, pc, act=ops.ACT_RELU, res=res, res_mode=ops.RES_SAME, algo=algo, out=y)
        b.record()
        torch.cuda.synchronize()
        t = a.elapsed_time(b) / 10
        out.append((t, algo))
    out.sort()
    print("M=%d %d->%d: %s" % (n * h * w, cin, cout, "  ".join("algo%d %.1fus %.2fTB/s" % (al, t * 1e3, byts / t / 1e9) for t, al in out[:4])), flush=True)
