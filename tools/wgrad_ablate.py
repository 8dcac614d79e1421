import os, sys
sys.path.insert(0, "/root/repo" if os.path.exists("/root/repo/oneshotdet_amd") else os.getcwd())
import torch
from oneshotdet_amd import ops
for (n, h, w, cin, cout, k, s, p) in [(8, 100, 128, 256, 256, 3, 1, 1), (8, 50, 64, 256, 256, 3, 1, 1)]:
    x = torch.randn(n, h, w, cin, device="cuda").bfloat16()
    ho, wo = ops.conv_out(h, k, s, p), ops.conv_out(w, k, s, p)
    dy = torch.randn(n, ho, wo, cout, device="cuda").bfloat16()
    dw = torch.zeros(cout, k, k, cin, device="cuda")
    fl = 2.0 * n * ho * wo * cout * cin * k * k
    out = []
    for algo in (1 + 0 + 16 * 0, 1 + 0 + 16 * 4, 1 + 4 + 16 * 1, 1 + 4 + 16 * 2, 1 + 4 + 16 * 0, 1 + 1 + 16 * 0, 1 + 3 + 16 * 0):
        ops.conv2d_wgrad(x, dy, dw, k, k, s, p, cout, algo=algo)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5):
            ops.conv2d_wgrad(x, dy, dw, k, k, s, p, cout, algo=algo)
        b.record()
        torch.cuda.synchronize()
        t = a.elapsed_time(b) / 5
        out.append("v%dt%d %.0fus %.0fTF" % ((algo - 1) & 7, (algo - 1) >> 3, t * 1e3, fl / t / 1e9))
    print("M=%d %dx%d->%d: %s" % (n * ho * wo, k, cin, cout, "  ".join(out)), flush=True)
