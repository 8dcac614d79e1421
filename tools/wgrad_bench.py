"""GPU box: time the weight-gradient kernel on representative shapes (bf16)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oneshotdet_amd import ops
dt = torch.bfloat16
shapes = [(8, 100, 128, 256, 256, 3, 1, 1), (8, 50, 64, 256, 256, 3, 1, 1), (8, 50, 64, 1024, 256, 1, 1, 0),
          (8, 25, 32, 512, 512, 3, 1, 1), (8, 100, 128, 128, 512, 1, 1, 0), (8, 100, 128, 512, 256, 1, 1, 0)]
for (n, h, w, cin, cout, k, s, p) in shapes:
    x = torch.randn(n, h, w, cin, device="cuda").to(dt)
    ho, wo = ops.conv_out(h, k, s, p), ops.conv_out(w, k, s, p)
    dy = torch.randn(n, ho, wo, cout, device="cuda").to(dt)
    dw = torch.zeros(cout, k, k, cin, device="cuda")
    fl = 2.0 * n * ho * wo * cout * cin * k * k
    ops.conv2d_wgrad(x, dy, dw, k, k, s, p, cout)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        ops.conv2d_wgrad(x, dy, dw, k, k, s, p, cout)
    b.record()
    torch.cuda.synchronize()
    t = a.elapsed_time(b) / 10
    print("M=%d N=%d K=%d: %.1f us %.0f TF" % (n * ho * wo, cout, cin * k * k, t * 1e3, fl / t / 1e9))
