"""GPU box: the row-reuse 3x3 kernel (tile id 6) against the LDS-DMA 256x256 kernel on the shapes it targets."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oneshotdet_amd import ops  # noqa: E402


def bench(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    torch.cuda._sleep(int(60e6))      # park the stream: the host needs ~100 us per grouped call, about what the kernel runs
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(reps):
        fn()
    ev[1].record()
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / reps * 1e3


def main():
    g = torch.Generator(device="cuda").manual_seed(0)
    algos = {"dma256/ring2": 1 + 8 + 4, "sp": 1 + 8 + 6}
    if os.environ.get("XR_ALGOS"):
        algos = {a: int(a) for a in os.environ["XR_ALGOS"].split(",")}
    for name, shapes, cin, cout in [("tower P3+P4 (grouped)", [(8, 100, 128), (8, 50, 64)], 256, 256),
                                    ("tower all levels (grouped)", [(8, 100, 128), (8, 50, 64), (8, 25, 32), (8, 13, 16), (8, 7, 8)], 256, 256),
                                    ("P3 only", [(8, 100, 128)], 256, 256), ("P4 only", [(8, 50, 64)], 256, 256),
                                    ("layer2 3x3 128->128", [(8, 100, 128)], 128, 128)]:
        xs = [torch.randn((n, h, w, cin), device="cuda", generator=g).bfloat16() for n, h, w in shapes]
        if os.environ.get("XR_RELU"):       # post-ReLU activations: half the operand is zero (what the tower convs see)
            xs = [torch.relu(x) for x in xs]
        wt = torch.randn((cout, cin, 3, 3), device="cuda", generator=g) / (cin * 9) ** 0.5
        pc = ops.pack_conv(wt, bias=torch.zeros(cout, device="cuda"), dtype=torch.bfloat16)
        flops = sum(2.0 * n * h * w * cout * cin * 9 for n, h, w in shapes)
        out = []
        times = {an: [] for an in algos}
        rounds = int(os.environ.get("XR_ROUNDS", "5"))
        for rnd_ in range(rounds):                # interleaved rounds in one process: medians, not single shots
            for an, a in algos.items():
                try:
                    if len(xs) == 1:
                        t = bench(lambda: ops.conv2d(xs[0], pc, pad=1, algo=a), reps=10)
                    else:
                        t = bench(lambda: ops.conv2d_grouped(xs, pc, pad=1, algo=a, _whole=True), reps=10)
                    times[an].append(t)
                except Exception as e:      # noqa: BLE001
                    pass
        for an in algos:
            ts = sorted(times[an])
            if ts:
                out.append("%s med %.1f min %.1f us %.0f TF" % (an, ts[len(ts) // 2], ts[0], flops / ts[len(ts) // 2] / 1e6))
            else:
                out.append("%s n/a" % an)
        print("%-28s %s" % (name, " | ".join(out)), flush=True)


if __name__ == "__main__":
    main()
