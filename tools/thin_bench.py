import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from oneshotdet_amd import ops, _lib
for (n, h, w, cin, cout) in [(8, 200, 256, 64, 64), (8, 100, 128, 64, 64)]:
    x = torch.relu(torch.randn(n, h, w, cin, device="cuda")).bfloat16()
    pc = ops.pack_conv(torch.randn(cout, cin, 3, 3, device="cuda") / 24, bias=torch.zeros(cout, device="cuda"), dtype=torch.bfloat16)
    res = []
    for algo in ops.conv_algo_candidates(cout, False):
        try:
            y = ops.conv2d(x, pc, pad=1, act=ops.ACT_RELU, algo=algo)
        except _lib.OsdError:
            continue
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            ops.conv2d(x, pc, pad=1, act=ops.ACT_RELU, algo=algo, out=y)
        b.record()
        torch.cuda.synchronize()
        res.append((a.elapsed_time(b) * 100, algo))
    res.sort()
    print((n, h, w, cin, cout), ["%d: %.1f us" % (a, t) for t, a in res[:6]], [("%d: %.1f" % (a, t)) for t, a in res if a == 15])
