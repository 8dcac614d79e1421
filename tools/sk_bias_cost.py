"""GPU box: what the fused bias gradient costs the tower weight-gradient launch (4 convs x 5 levels, bs 8, team mode = algo 4):
the launch with and without d bias, interleaved."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oneshotdet_amd import ops  # noqa: E402

algo = int(os.environ.get("SK_ALGO", "4"))
g = torch.Generator(device="cuda").manual_seed(0)
rnd = lambda *s: (torch.randn(*s, device="cuda", generator=g) * 0.5).to(torch.bfloat16)  # noqa: E731
levels = [(8, 100, 128), (8, 50, 64), (8, 25, 32), (8, 13, 16), (8, 7, 8)]
dws = [torch.zeros(256, 3, 3, 256, device="cuda") for _ in range(4)]
dbs = [torch.zeros(256, device="cuda") for _ in range(4)]
xs = [[rnd(n, h, w, 256) for (n, h, w) in levels] for _ in range(4)]
dys = [[rnd(n, h, w, 256) for (n, h, w) in levels] for _ in range(4)]
with_b = [(xs[i][l], dys[i][l], dws[i], None, dbs[i]) for i in range(4) for l in range(5)]
without = [(xs[i][l], dys[i][l], dws[i], None, None) for i in range(4) for l in range(5)]


def t(items, reps=10):
    ops.conv2d_wgrad_multi(items, 3, 3, 1, 1, 256, algo=algo)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.conv2d_wgrad_multi(items, 3, 3, 1, 1, 256, algo=algo)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for r in range(3):
    print("round %d: with d bias %.1f us, without %.1f us" % (r, t(with_b), t(without)), flush=True)
