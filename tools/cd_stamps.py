"""GPU box, diagnostic build (OSD_BUILD_TAG=cdstamps OSD_BUILD_FLAGS=-DOSD_CD_STAMPS, OSD_LIB_PATH=...): where a workgroup of
conv_dma_kernel spends its cycles on the HBM-bound 1x1 convs of the bottlenecks — operand prologue (first stage landed), K loop,
epilogue (residual / mask loads, arithmetic, stores issued), store drain — and how many workgroups a CU holds at a time.
s_memtime cycles per wave, medians over workgroups.   python tools/cd_stamps.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oneshotdet_amd import ops  # noqa: E402

# (n, h, w, cin, cout, residual, mask) x algorithm ids (1 + impl * 32 + variant * 8 + tile)
CASES = [((8, 50, 64, 256, 1024, True, False), "layer3 conv3 + residual"),
         ((8, 50, 64, 256, 1024, True, True), "layer3 conv1 data gradient + residual + mask"),
         ((8, 100, 128, 128, 512, True, False), "layer2 conv3 + residual"),
         ((8, 50, 64, 1024, 256, False, False), "layer3 conv1 (reducing)")]
WAVES = {0: 4, 1: 4, 2: 4, 4: 8, 7: 8}
TILE = {0: (128, 128), 1: (128, 64), 2: (64, 64), 4: (256, 256), 7: (256, 128)}


def main():
    g = torch.Generator(device="cuda").manual_seed(0)
    for (n, h, w, cin, cout, has_res, has_mask), label in CASES:
        x = torch.relu(torch.randn((n, h, w, cin), device="cuda", generator=g)).bfloat16()
        wt = torch.randn((cout, cin, 1, 1), device="cuda", generator=g) / cin ** 0.5
        pc = ops.pack_conv(wt, bias=torch.zeros(cout, device="cuda"), dtype=torch.bfloat16)
        res = torch.randn((n, h, w, cout), device="cuda", generator=g).bfloat16() if has_res else None
        mask = torch.randn((n, h, w, cout), device="cuda", generator=g).bfloat16() if has_mask else None
        m = n * h * w
        byts = m * cin * 2 + m * cout * 2 * (1 + int(has_res) + int(has_mask))
        print("== %s: M=%d N=%d K=%d, %.0f MB" % (label, m, cout, cin, byts / 1e6), flush=True)
        for variant in (1, 3, 0, 2):
            for tile in (0, 1, 4, 7, 2):
                algo = 1 + variant * 8 + tile
                bm, bn = TILE[tile]
                nw = WAVES[tile]
                tiles = ((m + bm - 1) // bm) * ((cout + bn - 1) // bn)
                buf = torch.zeros((tiles * nw * 8,), device="cuda", dtype=torch.int64)
                kw = dict(res=res, res_mode=ops.RES_SAME if has_res else ops.RES_NONE, mask=mask, act=ops.ACT_RELU if not has_mask else ops.ACT_NONE,
                          algo=algo, act_scale_dev=buf.view(torch.float32))
                try:
                    y = ops.conv2d(x, pc, **kw)
                except Exception as e:      # noqa: BLE001
                    continue
                for _ in range(3):
                    ops.conv2d(x, pc, out=y, **kw)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    ops.conv2d(x, pc, out=y, **kw)
                e1.record()
                torch.cuda.synchronize()
                us = e0.elapsed_time(e1) * 100
                t = buf.view(tiles, nw, 8).cpu().double()
                start, dur = t[:, 0, 0], t[:, :, 1:5].sum(-1).max(-1).values
                span = float((start + dur).max() - start.min())
                conc = float(dur.sum() / span / 256)
                med = [float(t[:, :, k].reshape(-1).median()) for k in range(1, 5)]
                print("  v%d tile%d (%3dx%3d, %d waves) %6.1f us %5.2f TB/s | cycles: prologue %6.0f  K loop %6.0f  epilogue %6.0f  store drain %6.0f | "
                      "workgroup %6.0f  launch span %7.0f  workgroups in flight per CU %.2f" %
                      (variant, tile, bm, bn, nw, us, byts / us / 1e6, med[0], med[1], med[2], med[3], float(dur.median()), span, conc), flush=True)


if __name__ == "__main__":
    main()
