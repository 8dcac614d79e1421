"""GPU box: race screen + timing of the ping-pong 256x256 conv kernel (algo tile 5) against the LDS-DMA 256x256 kernel
(tile 4), which accumulates in the same order: outputs must be BIT-IDENTICAL on every repeat."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oneshotdet_amd import ops

P8, DMA = 1 + 5, 1 + 8 + 4
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
torch.manual_seed(0)
bad = 0
for (n, h, w, cin, cout, k, pad) in [(8, 100, 128, 256, 256, 3, 1), (8, 50, 64, 256, 256, 3, 1), (2, 37, 41, 256, 256, 3, 1),
                                     (8, 50, 64, 1024, 256, 1, 0), (8, 50, 64, 256, 1024, 1, 0), (1, 9, 7, 64, 256, 3, 1),
                                     (8, 25, 32, 512, 512, 3, 1), (3, 13, 16, 128, 320, 1, 0), (8, 100, 128, 512, 256, 1, 0)]:
    x = torch.randn(n, h, w, cin, device="cuda").bfloat16()
    wt = torch.randn(cout, cin, k, k, device="cuda") / (cin * k * k) ** 0.5
    pc = ops.pack_conv(wt, bias=torch.randn(cout, device="cuda"), dtype=torch.bfloat16)
    res = torch.randn(n, h, w, pc.cout_store, device="cuda").bfloat16()
    ref = ops.conv2d(x, pc, pad=pad, res=res, res_mode=ops.RES_SAME, act=ops.ACT_RELU, algo=DMA)
    nbad = 0
    for _ in range(reps):
        y = ops.conv2d(x, pc, pad=pad, res=res, res_mode=ops.RES_SAME, act=ops.ACT_RELU, algo=P8)
        if not torch.equal(y, ref):
            nbad += 1
    torch.cuda.synchronize()
    fl = 2.0 * n * h * w * cout * cin * k * k
    tt = []
    for algo in (P8, DMA):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        y = ops.conv2d(x, pc, pad=pad, algo=algo)
        a.record()
        for _ in range(10):
            ops.conv2d(x, pc, pad=pad, algo=algo, out=y)
        b.record()
        torch.cuda.synchronize()
        tt.append(a.elapsed_time(b) / 10)
    bad += nbad
    print("n%d %dx%d cin%d cout%d k%d: mismatching repeats %d/%d   p8 %.1f us %.0f TF   dma256 %.1f us %.0f TF" % (
        n, h, w, cin, cout, k, nbad, reps, tt[0] * 1e3, fl / tt[0] / 1e9, tt[1] * 1e3, fl / tt[1] / 1e9), flush=True)
print("TOTAL MISMATCHES", bad)
