"""GPU box (diagnostic): the one-pass GroupNorm kernels on two streams at once."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oneshotdet_amd import ops  # noqa: E402

g = torch.Generator(device="cuda").manual_seed(3)
sizes = [(100, 128), (50, 64), (25, 32), (13, 16), (7, 8)]
n, c = 4, 256
xs = [(torch.randn((n, h, w, c), device="cuda", generator=g) * 2 + 0.3).bfloat16() for h, w in sizes]
dts = [torch.randn((n, h, w, c), device="cuda", generator=g).bfloat16() for h, w in sizes]
gamma = torch.randn(c, device="cuda", generator=g)
beta = torch.randn(c, device="cuda", generator=g)
mode = sys.argv[1] if len(sys.argv) > 1 else "both"


def run():
    ys, ab = ops.groupnorm_relu_levels(xs, gamma, beta, 32, 1e-5)
    if mode == "fwd":
        return ys, ab, None
    dg, db = torch.zeros(c, device="cuda"), torch.zeros(c, device="cuda")
    dus = ops.groupnorm_relu_bwd_levels(xs, dts, ab, gamma, beta, dg, db, 32)
    return ys, ab, dus


y0, ab0, du0 = run()
torch.cuda.synchronize()
print("quiet run done", flush=True)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
for s in (s1, s2):
    s.wait_stream(torch.cuda.current_stream())
for rep in range(10):
    outs = {}
    for s in (s1, s2):
        with torch.cuda.stream(s):
            outs[s] = run()
    torch.cuda.synchronize()
    ok = all(torch.equal(p, q) for s in (s1, s2) for p, q in zip(outs[s][0], y0))
    print("rep", rep, "equal", ok, "errors", ops.gn_onepass_errors(), flush=True)
