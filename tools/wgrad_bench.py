"""GPU box: time every weight-gradient algorithm (variant x split target) on the big shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oneshotdet_amd import ops

shapes = [(8, 100, 128, 256, 256, 3, 1, 1), (8, 50, 64, 256, 256, 3, 1, 1), (8, 50, 64, 1024, 256, 1, 1, 0), (8, 25, 32, 512, 512, 3, 1, 1)]
for (n, h, w, cin, cout, k, s, p) in shapes:
    x = torch.randn(n, h, w, cin, device="cuda").bfloat16()
    ho, wo = ops.conv_out(h, k, s, p), ops.conv_out(w, k, s, p)
    dy = torch.randn(n, ho, wo, cout, device="cuda").bfloat16()
    dw = torch.zeros(cout, k, k, cin, device="cuda")
    fl = 2.0 * n * ho * wo * cout * cin * k * k
    res = []
    for algo in ops.wgrad_algo_candidates(ops.OSD_BF16, cout, cin) + [1 + 4 + 16 * 5, 1 + 4 + 16 * 6]:
        ops.conv2d_wgrad(x, dy, dw, k, k, s, p, cout, algo=algo)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5):
            ops.conv2d_wgrad(x, dy, dw, k, k, s, p, cout, algo=algo)
        b.record()
        torch.cuda.synchronize()
        t = a.elapsed_time(b) / 5
        res.append((t, "v%d t%d" % ((algo - 1) & 7, (algo - 1) >> 3)))
    best = sorted(res)[:4]
    v4 = sorted(r for r in res if r[1].startswith("v4"))[:3]
    print("M=%d %dx%d->%d: best %s | 256-tile %s" % (n * ho * wo, k, cin, cout,
          "  ".join("%s %.0fus %.0fTF" % (nm, t * 1e3, fl / t / 1e9) for t, nm in best),
          "  ".join("%s %.0fus %.0fTF" % (nm, t * 1e3, fl / t / 1e9) for t, nm in v4)), flush=True)
