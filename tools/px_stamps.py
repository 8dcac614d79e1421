"""GPU box, diagnostic build (OSD_BUILD_TAG=pxstamps OSD_BUILD_FLAGS=-DOSD_PX_STAMPS, OSD_LIB_PATH=...): where a workgroup of
conv_px_kernel spends its cycles, per work unit: waiting for the weight chunk, at the barrier, issuing the next unit's fetches,
MFMAs, waiting for the unit's operands, epilogue.  s_memtime cycles per wave, medians over workgroups.   python tools/px_stamps.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oneshotdet_amd import ops  # noqa: E402

CASES = [(8, 50, 64, 256, 1024, True, False), (8, 50, 64, 256, 1024, True, True), (8, 100, 128, 128, 512, True, False),
         (8, 100, 128, 128, 512, True, True), (8, 200, 256, 64, 256, True, False)]
g = torch.Generator(device="cuda").manual_seed(0)
for n, h, w, cin, cout, has_res, has_mask in CASES:
    x = torch.relu(torch.randn((n, h, w, cin), device="cuda", generator=g)).bfloat16()
    wt = torch.randn((cout, cin, 1, 1), device="cuda", generator=g) / cin ** 0.5
    pc = ops.pack_conv(wt, bias=torch.zeros(cout, device="cuda"), dtype=torch.bfloat16)
    res = torch.randn((n, h, w, cout), device="cuda", generator=g).bfloat16() if has_res else None
    mask = torch.randn((n, h, w, cout), device="cuda", generator=g).bfloat16() if has_mask else None
    grid = min(512, ((n * h * w + 127) // 128) * (cout // 64))
    buf = torch.zeros((grid * 8 * 10,), device="cuda", dtype=torch.int64)
    kw = dict(res=res, res_mode=ops.RES_SAME if has_res else ops.RES_NONE, mask=mask, act=ops.ACT_NONE if has_mask else ops.ACT_RELU,
              algo=ops.CONV_ALGO_PX_WIDE if os.environ.get('OSD_PX_WIDE') else ops.CONV_ALGO_PX, act_scale_dev=buf.view(torch.float32))
    y = ops.conv2d(x, pc, **kw)
    for _ in range(3):
        ops.conv2d(x, pc, out=y, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.conv2d(x, pc, out=y, **kw)
    e1.record()
    torch.cuda.synchronize()
    t = buf.view(grid, 8, 10).cpu().double()
    t = t[:, :(4 if os.environ.get('OSD_PX_WIDE') else 8)]
    units = t[:, :, 8]
    per = lambda k: float((t[:, :, k] / units).reshape(-1).median())      # noqa: E731
    span = float((t[:, :, 0] + t[:, :, 1]).max() - t[:, :, 0].min())
    cu = (t[:, 0, 9].long() & 0xff) | ((t[:, 0, 9].long() >> 8) << 8)
    print("M=%d N=%d K=%d res=%d mask=%d: %.1f us; per unit (cycles, wave medians): total %.0f = wait weights %.0f + barrier %.0f + issue %.0f + mfma %.0f + "
          "wait operands %.0f + epilogue %.0f; units per workgroup %.2f; launch span %.0f cycles, workgroup start spread %.0f, workgroup lifetime median %.0f / max %.0f"
          % (n * h * w, cout, cin, has_res, has_mask, e0.elapsed_time(e1) * 100, per(1), per(2), per(3), per(4), per(5), per(6), per(7), float(units.mean()), span,
             float(t[:, 0, 0].max() - t[:, 0, 0].min()), float(t[:, 0, 1].median()), float(t[:, 0, 1].max())), flush=True)
