"""GPU box: the tower data-gradient conv over P3 + P4 of both towers (bs 8) as a plain launch of the software-pipelined kernel and
with the GroupNorm-backward statistics gathered in its epilogue (osd_conv2d_fwd_multi_gn), and the GroupNorm backward of those
levels with / without its statistics pass.   python tools/gnb_bench.py"""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import torch
from oneshotdet_amd import ops

n, c, G = 8, 256, 32
g = torch.Generator(device="cuda").manual_seed(0)
rnd = lambda *s: (torch.randn(*s, device="cuda", generator=g) * 0.5)
sizes = [(100, 128), (50, 64)]
pcs = [ops.pack_conv(rnd(c, c, 3, 3) / 48, bias=torch.zeros(c, device="cuda"), dtype=torch.bfloat16) for _ in range(2)]
gam = [rnd(c) for _ in range(2)]
bet = [rnd(c) for _ in range(2)]
seg_x, seg_u, seg_pc = [], [], []
for (h, w) in sizes:
    for tw in range(2):
        seg_x.append(rnd(n, h, w, c).bfloat16()); seg_u.append((rnd(n, h, w, c) * 2 + 0.3).bfloat16()); seg_pc.append(pcs[tw])
abs_ = [ops.groupnorm_relu_levels([seg_u[2 * l + tw] for l in range(2)], gam[tw], bet[tw], G, 1e-5)[1] for tw in range(2)]
wss = [torch.zeros(ops.gn_bwd_ws_numel(2, n, c, G), device="cuda") for _ in range(2)]
gnb = {"us": [], "abs": [], "gammas": [], "wss": [], "pws": [], "n": n, "groups": G}
for l in range(2):
    for tw in range(2):
        ws_l, pw_l = ops.gn_bwd_ws_parts(wss[tw], 2, n, c, G)[l]
        gnb["us"].append(seg_u[2 * l + tw]); gnb["abs"].append(abs_[tw][l]); gnb["gammas"].append(gam[tw])
        gnb["wss"].append(ws_l); gnb["pws"].append(pw_l)


def t(fn, reps=10):
    fn(); torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(reps):
        fn()
    ev[1].record(); torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / reps * 1e3


dts = ops.conv2d_multi(seg_x, seg_pc, pad=1, algo=ops.ALGO_SP, _whole=True)
dg, db = torch.zeros(c, device="cuda"), torch.zeros(c, device="cuda")
for r in range(3):
    a = t(lambda: ops.conv2d_multi(seg_x, seg_pc, pad=1, algo=ops.ALGO_SP, _whole=True))
    b = t(lambda: ops.conv2d_multi(seg_x, seg_pc, pad=1, gnb=gnb))
    c0 = t(lambda: ops.groupnorm_relu_bwd_levels([seg_u[0], seg_u[2]], [dts[0], dts[2]], abs_[0], gam[0], bet[0], dg, db, G))
    c1 = t(lambda: ops.groupnorm_relu_bwd_levels([seg_u[0], seg_u[2]], [dts[0], dts[2]], abs_[0], gam[0], bet[0], dg, db, G, ws=wss[0], fused_mask=3))
    z = t(lambda: wss[0].zero_())
    fw = [torch.zeros(2 * n * ops.GN_SPLITS * G * 2, device="cuda") for _ in range(2)]
    fparts = [ops.gn_fwd_ws_parts(fw[tw], 2, n, G) for tw in range(2)]
    gnf = {"wss": [fparts[tw][l] for l in range(2) for tw in range(2)], "n": n, "groups": G}
    f = t(lambda: ops.conv2d_multi(seg_x, seg_pc, pad=1, gnb=gnf))
    g0 = t(lambda: ops.groupnorm_relu_levels([dts[0], dts[2]], gam[0], bet[0], G, 1e-5))
    g1 = t(lambda: ops.groupnorm_relu_levels([dts[0], dts[2]], gam[0], bet[0], G, 1e-5, ws=fw[0], fused_mask=3))
    print("conv + FORWARD statistics %.1f us   GN forward (one tower) two passes %.1f us   apply only %.1f us" % (f, g0, g1))
    print("conv plain %.1f us   conv + statistics %.1f us   GN backward (one tower) two passes %.1f us   apply only %.1f us   zeroing one workspace %.1f us"
          % (a, b, c0, c1, z))
