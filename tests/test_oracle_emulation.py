"""CPU: pins of the two reduced-precision instruments the bf16 parity tests stand on.

1. `oracle.hotpath_ref.Emulation` (the oracle's bf16 mode): with the rounding switched off it IS the fp32 restatement (only
   the FrozenBN fold differs), forward and backward; with it on it differs from fp32 by what 8-bit storage costs.
2. The sensitivity of that rounded pipeline: a 1e-6 relative perturbation of the weights — what a different fp32 summation
   order amounts to — moves its outputs and gradients as far as bf16 is from fp32.  This is the floor under ANY end-to-end
   bf16 comparison (engine vs emulation included) and the reason tests/test_gpu_launch_replay.py checks launch by launch.
3. `oracle.launch_replay`: each launch restatement against the reference's own expressions / torch autograd.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import golden_utils as gu
from oneshotdet_amd import spec, synth
from oracle import hotpath_ref as orc
from oracle import launch_replay as lr


def _run(emu, name="small", weight_noise=0.0):
    B, H, W, S, qh, qw = gu.CASES[name]
    img, q = gu.case_inputs(name)
    sd = orc.to_torch_state_dict(synth.make_state_dict(spec.hot_path_shapes()))
    if weight_noise:
        g = torch.Generator().manual_seed(1)
        for k, v in sd.items():
            if v.dim() == 4:
                v.mul_(1.0 + weight_noise * torch.randn(v.shape, generator=g))
    for k, v in sd.items():
        v.requires_grad_(not spec.is_frozen(k))
    o = orc.hot_path_forward(torch.from_numpy(img), torch.from_numpy(q), sd, shots=S, emu=emu)
    gts = synth.make_gt_boxes(B, H, W, seed=3, max_boxes=3)
    c, r, t, info = orc.fcos_loss(o["logits"], o["bbox_reg"], o["centerness"], gts, focal="cuda")
    (c + r + t).backward()
    return o, np.array([c.item(), r.item(), t.item()]), {k: v.grad for k, v in sd.items() if v.grad is not None}


def _rel(a, b):
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


@pytest.fixture(scope="module")
def runs():
    torch.set_num_threads(8)
    return {"fp32": _run(None), "fold": _run(orc.Emulation(None)), "bf16": _run(orc.Emulation(torch.bfloat16)),
            "bf16_noise": _run(orc.Emulation(torch.bfloat16), weight_noise=1e-6)}


def test_emulation_with_rounding_off_is_the_fp32_oracle(runs):
    (o0, l0, g0), (o1, l1, g1) = runs["fp32"], runs["fold"]
    for key in ("features", "query_features", "combined", "logits", "bbox_reg", "centerness"):
        for a, b in zip(o1[key], o0[key]):
            assert _rel(a.detach(), b.detach()) <= 1e-5, key
    np.testing.assert_allclose(l1, l0, rtol=1e-5)
    assert set(g1) == set(g0)
    # gradients: every element within 3e-2 of the tensor's largest entry, relative L2 <= 5e-3 (FCOS's sparse gradients make a few tensors
    # ill-conditioned against their own norm: one ReLU on the edge moves a handful of entries), cosine >= 0.9999
    for k in g0:
        scale = float(g0[k].abs().max())
        if scale == 0.0:
            assert float(g1[k].abs().max()) == 0.0, k
            continue
        err = (g1[k] - g0[k]).abs() / max(scale, 1e-30)
        assert float(err.max()) <= 3e-2 and _rel(g1[k], g0[k]) <= 5e-3, (k, float(err.max()), _rel(g1[k], g0[k]))
        assert float((g1[k] * g0[k]).sum() / (g1[k].norm() * g0[k].norm()).clamp_min(1e-30)) >= 0.9999, k


def test_rounding_costs_what_8_bit_storage_costs(runs):
    """bf16 emulation vs fp32: features within 2e-2 relative L2 per level (measured 6e-3..9e-3), logits within 5e-2 RMS."""
    (o0, l0, g0), (o2, l2, g2) = runs["fp32"], runs["bf16"]
    for lvl in range(5):
        assert 1e-3 <= _rel(o2["features"][lvl].detach(), o0["features"][lvl].detach()) <= 2e-2
        assert float((o2["logits"][lvl] - o0["logits"][lvl]).detach().pow(2).mean().sqrt()) <= 5e-2
    np.testing.assert_allclose(l2, l0, rtol=3e-2)
    # every stored tensor of the emulation is exactly representable in bf16
    for key in ("features", "combined", "logits", "bbox_reg", "centerness"):
        for t in o2[key]:
            assert torch.equal(t.detach().bfloat16().float(), t.detach())


def test_bf16_pipeline_sensitivity_sets_the_end_to_end_floor(runs):
    """Two runs of the SAME rounded pipeline whose weights differ by 1e-6 relative (a stand-in for another fp32 summation
    order) end up as far apart as bf16 is from fp32: a boundary case of one rounding becomes a one-ulp (0.4 %) difference, the
    next layers' roundings amplify it, and after a few layers the two runs' rounding errors are independent.  So no end-to-end
    bf16 tolerance can be much tighter than the bf16-vs-fp32 one — measured here, asserted as a floor AND a ceiling."""
    (o0, _, g0), (o2, _, g2), (o3, _, g3) = runs["fp32"], runs["bf16"], runs["bf16_noise"]
    feat_same = [_rel(a.detach(), b.detach()) for a, b in zip(o3["features"], o2["features"])]
    feat_fp32 = [_rel(a.detach(), b.detach()) for a, b in zip(o2["features"], o0["features"])]
    assert all(2e-3 <= s <= 2e-2 for s in feat_same), feat_same
    assert all(0.5 <= s / f <= 2.0 for s, f in zip(feat_same, feat_fp32)), (feat_same, feat_fp32)
    keys = [k for k in g0 if not k.endswith(".scale")]
    same = sorted(_rel(g3[k], g2[k]) for k in keys)
    fp32 = sorted(_rel(g2[k], g0[k]) for k in keys)
    assert 0.05 <= same[len(same) // 2] <= 0.35 and same[-1] <= 0.6, (same[len(same) // 2], same[-1])
    assert 0.5 <= same[len(same) // 2] / fp32[len(fp32) // 2] <= 2.0


# ------------------------------------------------------------------------------------------------ launch restatements
def _pack(w_oihw, scale=None, rows=None, dtype=torch.float32):
    """What osd_pack_conv_weight produces: [rows][R][S][cin] with the FrozenBN scale folded into the rows."""
    w = w_oihw if scale is None else w_oihw * scale.view(-1, 1, 1, 1)
    cout = w.shape[0]
    rows = cout if rows is None else rows
    out = torch.zeros((rows,) + tuple(w.permute(0, 2, 3, 1).shape[1:]))
    out[:cout] = w.permute(0, 2, 3, 1)
    return out.to(dtype)


def test_conv_launches_chain_into_the_reference_bottleneck_and_fpn_lateral():
    """resnet.py:295-315 with a downsample branch (stride in the 1x1s) and fpn.py:57-60, built from conv_launch calls the way
    the engines issue them (folded FrozenBN, residual / nearest-2x addend and ReLU in the epilogue), equal the oracle's
    functions; the two-source form (conv3 + downsample as ONE GEMM) equals the two-launch form."""
    torch.manual_seed(0)
    sd, p = {}, "b."
    for name, (co, ci, k) in {"conv1": (16, 32, 1), "conv2": (16, 16, 3), "conv3": (64, 16, 1), "downsample.0": (64, 32, 1)}.items():
        sd[p + name + ".weight"] = torch.randn(co, ci, k, k) / np.sqrt(ci * k * k)
    for bn, c in {"bn1": 16, "bn2": 16, "bn3": 64, "downsample.1": 64}.items():
        sd[p + bn + ".weight"], sd[p + bn + ".bias"] = torch.rand(c) + 0.5, torch.randn(c) * 0.1
        sd[p + bn + ".running_mean"], sd[p + bn + ".running_var"] = torch.randn(c) * 0.1, torch.rand(c) + 0.5
    x = torch.randn(2, 32, 12, 12)
    want = orc.bottleneck(x, sd, p, stride=2)

    def folded(conv, bn):
        scale = sd[p + bn + ".weight"] * sd[p + bn + ".running_var"].rsqrt()
        return _pack(sd[p + conv + ".weight"], scale), sd[p + bn + ".bias"] - sd[p + bn + ".running_mean"] * scale
    xh = x.permute(0, 2, 3, 1).contiguous()
    w1, b1 = folded("conv1", "bn1")
    w2, b2 = folded("conv2", "bn2")
    w3, b3 = folded("conv3", "bn3")
    wd, bd = folded("downsample.0", "downsample.1")
    o1 = lr.conv_launch(xh, w1, b1, 16, 1, 1, stride=2, act=lr.ACT_RELU)
    o2 = lr.conv_launch(o1, w2, b2, 16, 3, 3, pad=1, act=lr.ACT_RELU)
    ident = lr.conv_launch(xh, wd, bd, 64, 1, 1, stride=2)
    y = lr.conv_launch(o2, w3, b3, 64, 1, 1, act=lr.ACT_RELU, res=ident, res_mode=lr.RES_SAME)
    torch.testing.assert_close(y.permute(0, 3, 1, 2), want, rtol=1e-5, atol=1e-5)
    y2 = lr.conv_launch(o2, torch.cat([w3, wd], -1), b3 + bd, 64, 1, 1, act=lr.ACT_RELU, x2=xh, x2_stride=2)
    torch.testing.assert_close(y2, y, rtol=1e-5, atol=1e-5)
    y3 = lr.conv_launch(o2, w3, b3 + bd, 64, 1, 1, act=lr.ACT_RELU, x2=xh, x2_stride=2, w2=wd)
    torch.testing.assert_close(y3, y, rtol=1e-5, atol=1e-5)
    # the bottleneck computed on every other pixel only (layer1's last block: its output is read by layer2.0's stride-2 1x1 convs and by
    # nothing else): the 3x3 at stride 2, conv3 + the identity at (2 ho, 2 wo) = the even pixels of the full block
    xs1 = torch.randn(2, 12, 12, 64)
    w1s, w2s, w3s = torch.randn(16, 64, 1, 1) * 0.1, torch.randn(16, 16, 3, 3) * 0.1, torch.randn(64, 16, 1, 1) * 0.1
    z16, z64 = torch.zeros(16), torch.zeros(64)
    f1 = lr.conv_launch(xs1, _pack(w1s), z16, 16, 1, 1, act=lr.ACT_RELU)
    full = lr.conv_launch(lr.conv_launch(f1, _pack(w2s), z16, 16, 3, 3, pad=1, act=lr.ACT_RELU), _pack(w3s), z64, 64, 1, 1, act=lr.ACT_RELU,
                          res=xs1, res_mode=lr.RES_SAME)
    half = lr.conv_launch(lr.conv_launch(f1, _pack(w2s), z16, 16, 3, 3, stride=2, pad=1, act=lr.ACT_RELU), _pack(w3s), z64, 64, 1, 1,
                          act=lr.ACT_RELU, res=xs1, res_mode=lr.RES_DOWN2X)
    torch.testing.assert_close(half, full[:, ::2, ::2], rtol=1e-5, atol=1e-5)
    # the query branch's pooling of all levels = ROIAlign (1 x 1, whole-image boxes) + the mean over the shots per level, and its backward
    # = autograd of that (generalized_rcnn.py:20-52, 100-104)
    feats = [torch.randn(4, 8, 8, 8), torch.randn(4, 4, 4, 8)]
    rois = torch.tensor([[i, 0.0, 0.0, 50.0 + 3 * i, 60.0 - 2 * i] for i in range(4)])
    pooled = lr.query_pool_launch(feats, rois, [1 / 8, 1 / 16], 2, 2)
    assert [tuple(t.shape) for t in pooled] == [(2, 8), (2, 8)]
    fa = [f.clone().requires_grad_(True) for f in feats]
    want_p = [orc.roi_align(f.permute(0, 3, 1, 2), rois, sc, 1, 1, 2).view(2, 2, 8).mean(1) for f, sc in zip(fa, [1 / 8, 1 / 16])]
    for a, b in zip(pooled, want_p):
        torch.testing.assert_close(a, b.detach(), rtol=1e-5, atol=1e-6)
    dqs = [torch.randn(2, 8), torch.randn(2, 8)]
    sum((w * d).sum() for w, d in zip(want_p, dqs)).backward()
    got_g = lr.query_pool_bwd_launch(dqs, rois, [tuple(f.shape) for f in feats], [1 / 8, 1 / 16], 2, 2)
    for g, f in zip(got_g, fa):
        torch.testing.assert_close(g, f.grad, rtol=1e-5, atol=1e-6)
    # FPN lateral + top-down (fpn.py:57-60) and P7 = conv(relu(P6)) (fpn.py:98)
    wl, bl = torch.randn(8, 64, 1, 1) * 0.1, torch.randn(8)
    top = torch.randn(2, 3, 3, 8)
    lat = lr.conv_launch(y, _pack(wl), bl, 8, 1, 1, res=top, res_mode=lr.RES_UP2X)
    want_lat = F.conv2d(want, wl, bl) + F.interpolate(top.permute(0, 3, 1, 2), scale_factor=2, mode="nearest")
    torch.testing.assert_close(lat.permute(0, 3, 1, 2), want_lat, rtol=1e-5, atol=1e-5)
    w7 = torch.randn(8, 8, 3, 3) * 0.1
    p7 = lr.conv_launch(lat, _pack(w7), bl, 8, 3, 3, stride=2, pad=1, relu_in=True)
    torch.testing.assert_close(p7.permute(0, 3, 1, 2), F.conv2d(F.relu(want_lat), w7, bl, stride=2, padding=1), rtol=1e-5, atol=1e-5)


def test_stem_launch_equals_the_reference_stem_on_the_packed_image():
    """resnet.py:332-337: 7x7 / 2 / pad 3 conv on the image == the stem launch on the zero-padded NHWC4 image with the packed
    [rows][7][8 pixels x 4 channels] weights."""
    torch.manual_seed(1)
    x = torch.randn(1, 3, 20, 26)
    w = torch.randn(16, 3, 7, 7) * 0.1
    ho, wo = 10, 13
    hp, wp = max(2 * (ho - 1) + 7, 20 + 3), max(2 * (wo - 1) + 8, 26 + 3)
    wp += wp & 1
    img = lr.pack_image_launch(x, (1, hp, wp, 4), 3, 3)
    packed = torch.zeros(16, 7, 8, 4)
    packed[:, :, :7, :3] = w.permute(0, 2, 3, 1)
    y = lr.conv_launch(img, packed.view(16, 7, 32), torch.zeros(16), 16, 7, 1, stem=True, out_hw=(ho, wo))
    torch.testing.assert_close(y.permute(0, 3, 1, 2), F.conv2d(x, w, None, stride=2, padding=3), rtol=1e-5, atol=1e-5)


def test_backward_launches_equal_autograd():
    """Data gradient of a stride-2 3x3 conv = scatter2x + the flipped-weight stride-1 conv; nearest-2x backward; the weight
    gradient with a folded FrozenBN scale; mask = ReLU backward of the producer; correlation's two gradients."""
    torch.manual_seed(2)
    x = torch.randn(2, 8, 9, 11, requires_grad=True)
    w = (torch.randn(6, 8, 3, 3) * 0.2).requires_grad_(True)
    scale = torch.rand(6) + 0.5
    y = F.conv2d(x, w * scale.view(-1, 1, 1, 1), None, stride=2, padding=1)
    dy = torch.randn_like(y)
    y.backward(dy)
    wf = (w.detach() * scale.view(-1, 1, 1, 1))
    w_dgrad = wf.flip(2, 3).permute(1, 0, 2, 3).contiguous()          # [cin][cout][r][s], taps flipped
    z = lr.scatter2x_launch(dy.permute(0, 2, 3, 1), (9, 11))
    dx = lr.conv_launch(z, _pack(w_dgrad), torch.zeros(8), 8, 3, 3, pad=1)
    torch.testing.assert_close(dx.permute(0, 3, 1, 2), x.grad, rtol=1e-4, atol=1e-5)
    dw, db = lr.wgrad_launch(x.detach().permute(0, 2, 3, 1), dy.permute(0, 2, 3, 1), 3, 3, 2, 1, 6, scale=scale, want_bias=True)
    torch.testing.assert_close(dw.permute(0, 3, 1, 2), w.grad, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(db, dy.sum(dim=(0, 2, 3)), rtol=1e-4, atol=1e-5)
    t = torch.randn(1, 4, 3, 5, requires_grad=True)
    up = F.interpolate(t, scale_factor=2, mode="nearest")
    g = torch.randn_like(up)
    up.backward(g)
    torch.testing.assert_close(lr.upsample2x_bwd_launch(g.permute(0, 2, 3, 1)).permute(0, 3, 1, 2), t.grad)
    a = torch.randn(1, 3, 4, 8)
    m = torch.randn(1, 3, 4, 8)
    torch.testing.assert_close(lr.add_mask_launch(a, a, m), torch.where(m > 0, 2 * a, torch.zeros_like(a)))
    f = torch.randn(2, 5, 6, 16, requires_grad=True)
    qv = torch.randn(2, 16, requires_grad=True)
    c = lr.correlate_launch(f, qv)
    gg = torch.randn_like(c)
    c.backward(gg)
    torch.testing.assert_close(lr.correlate_launch(gg, qv.detach()), f.grad)
    torch.testing.assert_close(lr.correlate_bwd_query_launch(gg, f.detach()), qv.grad, rtol=1e-5, atol=1e-5)


def test_loss_gradient_launch_equals_autograd_through_the_exp_scale_head():
    """fcos.py:95-97 + fcos/loss.py:213-276: with reg = exp(scale * x) stored, the launch's d/dx = d/dreg * reg * scale and
    d/dscale = sum d/dreg * reg * log(reg) / scale equal autograd through exp(scale * x)."""
    torch.manual_seed(3)
    sizes, scales = [(4, 5), (2, 3)], [1.3, 0.7]
    xs = [(torch.randn(1, 4, h, w) * 0.3 + 3.0).requires_grad_(True) for h, w in sizes]
    sc = [torch.tensor(s, requires_grad=True) for s in scales]
    logits = [torch.randn(1, 1, h, w).requires_grad_(True) for h, w in sizes]
    ctrs = [torch.randn(1, 1, h, w).requires_grad_(True) for h, w in sizes]
    gts = [np.array([[4.0, 6.0, 30.0, 28.0]], np.float32)]
    regs = [torch.exp(x * s) for x, s in zip(xs, sc)]
    c, r, t, info = orc.fcos_loss(logits, regs, ctrs, gts, focal="cuda")
    assert info["num_pos"] > 0
    (c + r + t).backward()
    head = [(torch.cat([lg, ct, torch.zeros_like(lg), torch.zeros_like(lg)], 1).detach().permute(0, 2, 3, 1),
             rg.detach().permute(0, 2, 3, 1)) for lg, ct, rg in zip(logits, ctrs, regs)]
    outs, raws, _ = lr.fcos_loss_grad_launch(head, torch.from_numpy(gts[0])[None], torch.tensor([1]), scales, 2.0, 0.25)
    for lvl in range(2):
        torch.testing.assert_close(outs[lvl][0][..., 0:1].permute(0, 3, 1, 2), logits[lvl].grad, rtol=1e-4, atol=1e-7)
        torch.testing.assert_close(outs[lvl][0][..., 1:2].permute(0, 3, 1, 2), ctrs[lvl].grad, rtol=1e-4, atol=1e-7)
        torch.testing.assert_close(outs[lvl][1].permute(0, 3, 1, 2), xs[lvl].grad, rtol=1e-4, atol=1e-7)
        assert abs(raws[lvl] / scales[lvl] - float(sc[lvl].grad)) <= 1e-4 * abs(float(sc[lvl].grad)) + 1e-7


def test_compare_counts_units_in_the_last_place():
    ref = torch.tensor([1.0, 1.00390625, -3.0, 1e-3, 100.0])
    good = ref.bfloat16()
    assert lr.compare(good, ref, torch.bfloat16, noise=0.0)["worst_ulp"] == 0.0
    one_up = torch.tensor([1.0078125, 1.0, -3.015625, 1e-3, 100.5]).bfloat16()           # each one bf16 step away
    res = lr.compare(one_up, ref, torch.bfloat16, noise=0.0)
    assert res["ok"] and 0.5 <= res["worst_ulp"] <= 1.0 and res["flips"] >= 0.6
    two_up = torch.tensor([1.015625, 1.0, -3.0, 1e-3, 100.0]).bfloat16()                  # two steps: not a rounding neighbour
    assert not lr.compare(two_up, ref, torch.bfloat16, noise=0.0)["ok"]
    assert not lr.compare(torch.tensor([float("nan")] * 5).bfloat16(), ref, torch.bfloat16)["ok"]
