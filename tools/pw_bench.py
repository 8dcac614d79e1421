"""GPU box: the bottleneck 1x1 convs of the bs=8 training step (resnet.py:295-315; forward with residual + ReLU, data gradient with
the block's residual gradient + ReLU mask) on the pixel-stationary pointwise kernel (conv_px, algos 49 / 50) against the best of the
one-tile-per-workgroup algorithms — time, algorithmic HBM bytes per second.  (Round 4's persistent conv_pw, algo 41, was retired in
round 5.)   python tools/pw_bench.py [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oneshotdet_amd import _lib, ops  # noqa: E402

REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 20
# (n, h, w, cin, cout, residual, mask, launches per step, what)
SHAPES = [(8, 50, 64, 256, 1024, True, False, 6, "layer3 conv3 + residual + ReLU"),
          (8, 50, 64, 256, 1024, True, True, 6, "layer3 conv1 data gradient + residual gradient + mask"),
          (8, 50, 64, 1024, 256, False, False, 6, "layer3 conv1 + ReLU"),
          (8, 50, 64, 1024, 256, False, True, 6, "layer3 conv3 data gradient + mask"),
          (8, 100, 128, 128, 512, True, False, 4, "layer2 conv3 + residual + ReLU"),
          (8, 100, 128, 128, 512, True, True, 3, "layer2 conv1 data gradient + residual gradient + mask"),
          (8, 100, 128, 512, 128, False, False, 4, "layer2 conv1 + ReLU"),
          (8, 100, 128, 512, 128, False, True, 3, "layer2 conv3 data gradient + mask"),
          (8, 25, 32, 512, 2048, True, False, 3, "layer4 conv3 + residual + ReLU"),
          (8, 25, 32, 2048, 512, False, False, 3, "layer4 conv1 + ReLU"),
          (8, 200, 256, 64, 256, True, False, 2, "layer1 conv3 + residual + ReLU"),
          (8, 200, 256, 256, 64, False, False, 2, "layer1 conv1 + ReLU"),
          (8, 100, 128, 512, 256, False, False, 1, "FPN lateral C3"),
          (8, 50, 64, 1024, 256, False, False, 1, "FPN lateral C4")]


def timed(fn):
    fn()
    torch.cuda.synchronize()
    best = float("inf")
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(REPS):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / REPS * 1e3)
    return best


def main():
    g = torch.Generator(device="cuda").manual_seed(0)
    tot_old = tot_new = 0.0
    print("| conv | M | N | K | MB | best one-tile algorithm | us | TB/s | (conv_pw: retired) | - | - | conv_px us | TB/s | launches/step |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|---|---|")
    for n, h, w, cin, cout, has_res, has_mask, per_step, what in SHAPES:
        x = torch.relu(torch.randn((n, h, w, cin), device="cuda", generator=g)).bfloat16()
        wt = torch.randn((cout, cin, 1, 1), device="cuda", generator=g) / cin ** 0.5
        pc = ops.pack_conv(wt, bias=torch.zeros(cout, device="cuda"), dtype=torch.bfloat16)
        res = torch.randn((n, h, w, cout), device="cuda", generator=g).bfloat16() if has_res else None
        mask = torch.randn((n, h, w, cout), device="cuda", generator=g).bfloat16() if has_mask else None
        y = torch.empty((n, h, w, cout), device="cuda", dtype=torch.bfloat16)
        kw = dict(res=res, res_mode=ops.RES_SAME if has_res else ops.RES_NONE, mask=mask, act=ops.ACT_NONE if has_mask else ops.ACT_RELU, out=y)
        m = n * h * w
        mb = (m * cin + cout * cin + m * cout * (1 + int(has_res) + int(has_mask))) * 2 / 1e6
        best, best_algo = float("inf"), 0
        for algo in ops.conv_algo_candidates(cout, False, has_mask=has_mask):
            if algo in (ops.CONV_ALGO_PX, ops.CONV_ALGO_PX_WIDE):
                continue
            try:
                ops.conv2d(x, pc, algo=algo, **kw)
            except _lib.OsdError:
                continue
            t = timed(lambda: ops.conv2d(x, pc, algo=algo, **kw))
            if t < best:
                best, best_algo = t, algo
        t_pw = float("nan")
        t_px = float("nan")
        for a_px in (ops.CONV_ALGO_PX, ops.CONV_ALGO_PX_WIDE):      # the faster of the two wave shapes
            try:
                ops.conv2d(x, pc, algo=a_px, **kw)
                t1 = timed(lambda: ops.conv2d(x, pc, algo=a_px, **kw))
                t_px = t1 if not (t_px == t_px) else min(t_px, t1)
            except _lib.OsdError:
                pass
        a0 = best_algo - 1
        fl = 2.0 * m * cin * cout
        print("| %s | %d | %d | %d | %.0f | impl %d variant %d tile %d | %.1f | %.2f | %.1f | %.2f | %.0f | %.1f | %.2f | %d |" %
              (what, m, cout, cin, mb, a0 >> 5, (a0 >> 3) & 3, a0 & 7, best, mb / best, t_pw, mb / t_pw, fl / t_pw / 1e6, t_px, mb / t_px, per_step), flush=True)
        tot_old += best * per_step
        tot_new += min([t for t in (best, t_pw, t_px) if t == t]) * per_step
    print("\nper step (one backbone): %.0f us with the one-tile algorithms, %.0f us with conv_px where it wins" % (tot_old, tot_new))


if __name__ == "__main__":
    main()
