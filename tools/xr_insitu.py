"""GPU box: why do all big-3x3 kernels take the same time inside the step?  Times the tower P3+P4 grouped conv per launch
(HIP events, stream parked behind a spin kernel so the host is out of the picture) in three settings:
  back-to-back | after a 512 MB streaming copy (caches flushed, clock relaxed) | after a GroupNorm-sized elementwise pass."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oneshotdet_amd import ops  # noqa: E402


def run(fn, pre, reps=12):
    evs = []
    torch.cuda.synchronize()
    torch.cuda._sleep(int(80e6))
    for _ in range(reps):
        if pre is not None:
            pre()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        evs.append((e0, e1))
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) * 1e3 for a, b in evs[2:])
    return t[len(t) // 2], t[0], t[-1]


def main():
    g = torch.Generator(device="cuda").manual_seed(0)
    shapes = [(8, 100, 128), (8, 50, 64)]
    cin = cout = 256
    xs = [torch.relu(torch.randn((n, h, w, cin), device="cuda", generator=g)).bfloat16() for n, h, w in shapes]
    wt = torch.randn((cout, cin, 3, 3), device="cuda", generator=g) / (cin * 9) ** 0.5
    pc = ops.pack_conv(wt, bias=torch.zeros(cout, device="cuda"), dtype=torch.bfloat16)
    outs = [torch.empty((n, h, w, cout), device="cuda", dtype=torch.bfloat16) for n, h, w in shapes]
    big_a = torch.empty(256 << 20, device="cuda", dtype=torch.uint8)
    big_b = torch.empty(256 << 20, device="cuda", dtype=torch.uint8)
    med = torch.randn((8, 100, 128, 256), device="cuda").bfloat16()
    flops = sum(2.0 * n * h * w * cout * cin * 9 for n, h, w in shapes)
    pres = {"back-to-back": None, "after 512 MB copy": lambda: big_b.copy_(big_a), "after 130 MB relu pass": lambda: torch.relu_(med)}
    for an, a in {"dma256/ring2": 1 + 8 + 4, "p8": 1 + 5, "xr": 1 + 6}.items():
        line = []
        for pn, pre in pres.items():
            med_t, lo, hi = run(lambda: ops.conv2d_grouped(xs, pc, pad=1, algo=a), pre)
            line.append("%s %.1f us (%.0f TF) [%.1f..%.1f]" % (pn, med_t, flops / med_t / 1e6, lo, hi))
        print("%-13s %s" % (an, " | ".join(line)), flush=True)


if __name__ == "__main__":
    main()
