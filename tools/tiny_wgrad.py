import os, sys
sys.path.insert(0, os.getcwd())
import torch
from oneshotdet_amd import ops
for (n, h, w, cin, cout, k, p) in [(8, 4, 4, 256, 256, 3, 1), (8, 8, 8, 256, 1024, 1, 0), (8, 1, 1, 256, 256, 3, 1)]:
    x = torch.randn(n, h, w, cin, device="cuda").bfloat16()
    dy = torch.randn(n, h, w, cout, device="cuda").bfloat16()
    dw = torch.zeros(cout, k, k, cin, device="cuda")
    scale = torch.ones(cout, device="cuda")
    for algo, sc in ((1, None), (1, scale), (1 + 16 * 7, scale)):
        ops.conv2d_wgrad(x, dy, dw, k, k, 1, p, cout, algo=algo, scale=sc)
        torch.cuda.synchronize()
        ts = []
        for reps in (1, 20):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda._sleep(int(3e9))          # park the stream: the host enqueues everything while the GPU waits
            a.record()
            for _ in range(reps):
                ops.conv2d_wgrad(x, dy, dw, k, k, 1, p, cout, algo=algo, scale=sc)
            b.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b) / reps * 1e3)
        print("M=%d %dx%d %d->%d algo %d scale %s: single %.1f us, back-to-back x20 %.1f us each" % (n * h * w, k, k, cin, cout, algo, sc is not None, ts[0], ts[1]))
