"""GPU box: time every conv algorithm on given shapes.  usage: conv_algo_bench.py [bf16|f32]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oneshotdet_amd import ops, _lib
dt = torch.float32 if (len(sys.argv) > 1 and sys.argv[1] == "f32") else torch.bfloat16
shapes = [(8, 100, 128, 256, 256, 3, 1, 1), (8, 50, 64, 256, 256, 3, 1, 1), (8, 50, 64, 1024, 256, 1, 1, 0),
          (8, 50, 64, 256, 1024, 1, 1, 0), (8, 25, 32, 512, 512, 3, 1, 1), (8, 200, 256, 64, 256, 1, 1, 0),
          (8, 25, 32, 256, 256, 3, 1, 1), (4, 128, 164, 256, 256, 3, 1, 1), (4, 80, 104, 256, 256, 3, 1, 1), (8, 100, 128, 128, 128, 3, 1, 1),
          (8, 100, 128, 128, 512, 1, 1, 0),
          # the latency-bound launches (query backbone, layer4, P6 / P7; 20 = config5's 4 x 5 queries)
          (8, 8, 8, 256, 256, 3, 1, 1), (8, 4, 4, 512, 512, 3, 1, 1), (8, 16, 16, 128, 128, 3, 1, 1), (20, 8, 8, 256, 256, 3, 1, 1),
          (8, 8, 8, 1024, 256, 1, 1, 0), (8, 4, 4, 2048, 512, 1, 1, 0), (8, 13, 16, 256, 256, 3, 1, 1)]
names = {0: "dma", 1: "reg"}
for (n, h, w, cin, cout, k, s, p) in shapes:
    x = torch.randn(n, h, w, cin, device="cuda").to(dt)
    wt = torch.randn(cout, cin, k, k, device="cuda") / (cin * k * k) ** 0.5
    pc = ops.pack_conv(wt, bias=torch.zeros(cout, device="cuda"), dtype=dt)
    ho, wo = ops.conv_out(h, k, s, p), ops.conv_out(w, k, s, p)
    fl = 2.0 * n * ho * wo * cout * cin * k * k
    res = []
    for algo in ops.conv_algo_candidates(cout, False):
        try:
            y = ops.conv2d(x, pc, stride=s, pad=p, algo=algo)
        except _lib.OsdError:
            continue
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            ops.conv2d(x, pc, stride=s, pad=p, algo=algo, out=y)
        b.record()
        torch.cuda.synchronize()
        t = a.elapsed_time(b) / 10
        a0 = algo - 1
        res.append((t, "%s v%d tile%d" % (names[a0 >> 5], (a0 >> 3) & 3, a0 & 7)))
    res.sort()
    print("M=%d N=%d K=%d:" % (n * ho * wo, cout, cin * k * k), "  ".join("%s %.1fus %.0fTF" % (nm, t * 1e3, fl / t / 1e9) for t, nm in res[:6]))
