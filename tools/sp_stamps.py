"""GPU box, diagnostic build (OSD_BUILD_TAG=stamps OSD_BUILD_FLAGS=-DOSD_SP_STAMPS, OSD_LIB_PATH=...): where a workgroup of
conv_sp_kernel spends its cycles — prologue, K loop, final barrier, epilogue, and inside the loop the three waits in front of
every stage barrier (fragment reads, LDS-DMA, the barrier itself).  s_memtime cycles per wave, medians over workgroups."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oneshotdet_amd import ops  # noqa: E402


def main():
    g = torch.Generator(device="cuda").manual_seed(0)
    n, h, w, cin, cout = 8, 100, 128, 256, 256
    x = torch.relu(torch.randn((n, h, w, cin), device="cuda", generator=g)).bfloat16()
    wt = torch.randn((cout, cin, 3, 3), device="cuda", generator=g) / (cin * 9) ** 0.5
    pc = ops.pack_conv(wt, bias=torch.zeros(cout, device="cuda"), dtype=torch.bfloat16)
    tiles = n * h * w // 256
    buf = torch.zeros((tiles * 8 * 8,), device="cuda", dtype=torch.int64)
    for _ in range(20):
        ops.conv2d(x, pc, pad=1, algo=15, act_scale_dev=buf.view(torch.float32))
    torch.cuda.synchronize()
    t = buf.view(tiles, 8, 8).cpu().double()
    names = ["start", "prologue", "K loop", "final barrier", "epilogue", "wait lgkmcnt (sum)", "wait vmcnt (sum)", "barrier (sum)"]
    start = t[:, :, 0]
    first = start.min()
    second_round = (start - first) > 0.4 * (start.max() - first)
    print("workgroups %d (second round: %d); wave-medians in cycles:" % (tiles, int(second_round[:, 0].sum())))
    for k in range(1, 8):
        v = t[:, :, k].reshape(-1)
        print("  %-22s median %8.0f  p10 %8.0f  p90 %8.0f" % (names[k], v.median(), v.quantile(0.1), v.quantile(0.9)))
    tot = t[:, :, 1:5].sum(-1).reshape(-1)
    print("  %-22s median %8.0f" % ("whole workgroup", tot.median()))
    print("  round 2 starts %.0f cycles after round 1 (median)" % float(start[second_round].median() - start[~second_round].median()))


if __name__ == "__main__":
    main()
