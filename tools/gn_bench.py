"""GPU box: isolated timing (stream parked) of the level-grouped GroupNorm+ReLU forward / backward at the FCOS tower
sizes (bs=8, P3..P7, 256 channels).  OSD_GN_APPLY_BLOCKS=<n> sets the apply kernels' workgroups per (image, level)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oneshotdet_amd import ops  # noqa: E402


def bench(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    torch.cuda._sleep(int(40e6))
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(reps):
        fn()
    ev[1].record()
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / reps * 1e3


g = torch.Generator(device="cuda").manual_seed(0)
sizes = ((100, 128), (50, 64), (25, 32), (13, 16), (7, 8))
xs = [torch.randn((8, h, w, 256), device="cuda", generator=g).bfloat16() for h, w in sizes]
dts = [torch.randn((8, h, w, 256), device="cuda", generator=g).bfloat16() for h, w in sizes]
gamma = torch.ones(256, device="cuda") * 1.1
beta = torch.zeros(256, device="cuda") + 0.05
ys, ab = ops.groupnorm_relu_levels(xs, gamma, beta, 32, 1e-5)
dg, db = torch.zeros(256, device="cuda"), torch.zeros(256, device="cuda")
mb = sum(x.numel() for x in xs) * 2 / 1e6
for onepass in (False, True):
    ops.GN_ONEPASS = ops.GN_ONEPASS_FWD = ops.GN_ONEPASS_BWD = onepass
    ys, ab = ops.groupnorm_relu_levels(xs, gamma, beta, 32, 1e-5)
    tf = bench(lambda: ops.groupnorm_relu_levels(xs, gamma, beta, 32, 1e-5))
    tb = bench(lambda: ops.groupnorm_relu_bwd_levels(xs, dts, ab, gamma, beta, dg, db, 32))
    # algorithmic bytes: forward reads u and writes t (2 x), backward reads u, dt and writes du (3 x); the two-launch forms move 3 x / 5 x
    print("%s (OSD_GN_APPLY_BLOCKS %s, OSD_GN1P_U_FWD %s, OSD_GN1P_U_BWD %s): forward %.1f us = %.2f TB/s of tensors; backward %.1f us = %.2f TB/s of tensors"
          % ("one pass  " if onepass else "two launch", os.environ.get("OSD_GN_APPLY_BLOCKS", "32"), os.environ.get("OSD_GN1P_U_FWD", "auto"),
             os.environ.get("OSD_GN1P_U_BWD", "8"), tf, 2 * mb / tf, tb, 3 * mb / tb), flush=True)
print("one-pass error words:", ops.gn_onepass_errors())
