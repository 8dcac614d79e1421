"""GPU (-m gpu): every HIP kernel through the C-ABI against the oracle / a plain PyTorch fp32 CPU reference of the
same op, on small seeded inputs incl. the edge cases (ragged M tails, odd sizes, empty inputs)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import golden_utils as gu
from oracle import hotpath_ref as orc

pytestmark = pytest.mark.gpu

DT = {"f32": torch.float32, "bf16": torch.bfloat16}
# fp32 path: exact-fp32 MFMA, only summation order differs from oneDNN.  bf16: 8-bit mantissa inputs, fp32 accumulate.
TOL = {"f32": dict(rtol=2e-5, atol=2e-5), "bf16": dict(rtol=3e-2, atol=3e-2)}


def ops():
    from oneshotdet_amd import ops as o
    return o


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).float()


def to_nhwc(x, dtype):
    return x.permute(0, 2, 3, 1).contiguous().to(dtype).cuda()


def from_nhwc(y):
    return y.float().cpu().permute(0, 3, 1, 2)


CONV_CASES = [
    # n, cin, h, w, cout, k, stride, pad
    (2, 64, 13, 17, 64, 1, 1, 0),
    (1, 128, 16, 12, 256, 1, 2, 0),
    (2, 64, 9, 11, 128, 3, 1, 1),
    (1, 256, 14, 10, 256, 3, 2, 1),
    (3, 64, 7, 5, 4, 3, 1, 1),          # skinny-N prediction conv
    (1, 512, 25, 32, 512, 3, 1, 1),     # deep K, 64x64 tiles
    (2, 256, 40, 48, 256, 3, 1, 1),     # 128x64 / 128x128 tiles with a ragged M tail
]


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv2d_bias_matches_torch(case, dt):
    n, cin, h, w, cout, k, s, p = case
    x, wt, b = rnd(n, cin, h, w, seed=1), rnd(cout, cin, k, k, seed=2) / np.sqrt(cin * k * k), rnd(cout, seed=3)
    if dt == "bf16":
        x, wt = x.bfloat16().float(), wt.bfloat16().float()
    ref = F.conv2d(x, wt, b, stride=s, padding=p)
    pc = ops().pack_conv(wt.cuda(), bias=b.cuda(), dtype=DT[dt])
    y = ops().conv2d(to_nhwc(x, DT[dt]), pc, stride=s, pad=p)
    assert y.shape[-1] == (cout + 3) // 4 * 4
    torch.testing.assert_close(from_nhwc(y)[:, :cout], ref, **TOL[dt])


@pytest.mark.parametrize("dt", ["f32", "bf16"])
def test_conv2d_frozenbn_residual_relu(dt):
    """Bottleneck tail: relu(bn3(conv3(x)) + identity) with the no-eps FrozenBN folded (batch_norm.py:19-24)."""
    n, cin, h, w, cout = 2, 64, 12, 20, 256
    x, wt, idn = rnd(n, cin, h, w, seed=1), rnd(cout, cin, 1, 1, seed=2) / 8, rnd(n, cout, h, w, seed=3)
    bn = [rnd(cout, seed=4).abs() + 0.5, rnd(cout, seed=5), rnd(cout, seed=6), rnd(cout, seed=7).abs() + 0.5]
    if dt == "bf16":
        x, idn = x.bfloat16().float(), idn.bfloat16().float()
    sd = {"bn.weight": bn[0], "bn.bias": bn[1], "bn.running_mean": bn[2], "bn.running_var": bn[3]}
    ref = F.relu(orc.frozen_bn(F.conv2d(x, wt), sd, "bn") + idn)
    pc = ops().pack_conv(wt.cuda(), bn=[t.cuda() for t in bn], dtype=DT[dt])
    y = ops().conv2d(to_nhwc(x, DT[dt]), pc, act=ops().ACT_RELU, res=to_nhwc(idn, DT[dt]), res_mode=ops().RES_SAME)
    torch.testing.assert_close(from_nhwc(y), ref, **TOL[dt])


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("stride", [1, 2])
def test_conv2d_two_sources_is_conv3_plus_downsample(stride, dt):
    """osd_conv2d_fwd with src2: relu(bn3(conv3(out)) + bn_d(downsample(x))) (resnet.py:295-315, first block of a stage;
    the downsample conv strides over the block input) as ONE 1x1 conv over the concatenated K — every algorithm the
    two-source form is built for, odd map sizes, channel counts of layer1.0 / layer3.0."""
    o = ops()
    for (cm, cx, cout, h, w) in ((64, 64, 256, 13, 21), (256, 512, 1024, 7, 9)):
        n = 2
        t2 = rnd(n, cm, h, w, seed=1)
        x = rnd(n, cx, (h - 1) * stride + 1 + (stride - 1), (w - 1) * stride + 1, seed=2)        # stride 2: odd / even extents
        w3, wd = rnd(cout, cm, 1, 1, seed=3) / cm ** 0.5, rnd(cout, cx, 1, 1, seed=4) / cx ** 0.5
        bn3 = [rnd(cout, seed=5).abs() + 0.5, rnd(cout, seed=6), rnd(cout, seed=7), rnd(cout, seed=8).abs() + 0.5]
        bnd = [rnd(cout, seed=9).abs() + 0.5, rnd(cout, seed=10), rnd(cout, seed=11), rnd(cout, seed=12).abs() + 0.5]
        if dt == "bf16":
            t2, x = t2.bfloat16().float(), x.bfloat16().float()
        sd = {"a.weight": bn3[0], "a.bias": bn3[1], "a.running_mean": bn3[2], "a.running_var": bn3[3],
              "b.weight": bnd[0], "b.bias": bnd[1], "b.running_mean": bnd[2], "b.running_var": bnd[3]}
        ref = F.relu(orc.frozen_bn(F.conv2d(t2, w3), sd, "a") + orc.frozen_bn(F.conv2d(x, wd, stride=stride), sd, "b"))
        s3, sdn = bn3[0] * bn3[3].rsqrt(), bnd[0] * bnd[3].rsqrt()
        wcat = torch.cat([w3 * s3.view(-1, 1, 1, 1), wd * sdn.view(-1, 1, 1, 1)], 1)
        bias = (bn3[1] - bn3[2] * s3) + (bnd[1] - bnd[2] * sdn)
        pc = o.pack_conv(wcat.cuda(), bias=bias.cuda(), dtype=DT[dt])
        tol = TOL[dt] if dt == "f32" else dict(rtol=3e-2, atol=3e-2)
        done = 0
        for algo in [None] + [1 + v * 8 + t for v in (0, 1, 2, 3) for t in (0, 2, 7)]:
            try:
                y = o.conv2d(to_nhwc(t2, DT[dt]), pc, act=o.ACT_RELU, x2=to_nhwc(x, DT[dt]), x2_stride=stride, algo=algo)
            except Exception as e:      # noqa: BLE001  (a tile that does not fit LDS with this ring)
                assert "does not fit" in str(e), (algo, e)
                continue
            torch.testing.assert_close(from_nhwc(y), ref, **tol)
            done += 1
        assert done >= 8
        # the same with two separately packed convs (their own FrozenBN folds) and the summed bias: what the training
        # engine uses for the trainable stages, whose conv3 / downsample weights are repacked every step
        pa = o.pack_conv(w3.cuda(), bn=[t.cuda() for t in bn3], dtype=DT[dt])
        pb = o.pack_conv(wd.cuda(), bn=[t.cuda() for t in bnd], dtype=DT[dt])
        for algo in (None, 1 + 0, 1 + 3 * 8 + 2, 1 + 2 * 8 + 7):
            y = o.conv2d(to_nhwc(t2, DT[dt]), pa, act=o.ACT_RELU, x2=to_nhwc(x, DT[dt]), x2_stride=stride, pc2=pb,
                         bias=(pa.bias + pb.bias), algo=algo)
            torch.testing.assert_close(from_nhwc(y), ref, **tol)
    with pytest.raises(Exception):       # 3x3 convs have no second source
        o.conv2d(to_nhwc(t2, DT[dt]), o.pack_conv(rnd(cout, cm + cx, 1, 1, seed=3).cuda(), bias=bias.cuda(), dtype=DT[dt]),
                 x2=to_nhwc(x[:, :, :2, :2], DT[dt]), x2_stride=stride)


@pytest.mark.parametrize("dt", ["f32", "bf16"])
def test_conv2d_fpn_lateral_upsample_add(dt):
    """inner = conv1x1(C) + nearest_up2x(top)  (fpn.py:59-64)."""
    n, cin, h, w, cout = 2, 128, 10, 14, 256
    x, wt, b, top = rnd(n, cin, h, w, seed=1), rnd(cout, cin, 1, 1, seed=2) / 11, rnd(cout, seed=3), rnd(n, cout, h // 2, w // 2, seed=4)
    if dt == "bf16":
        x, wt, top = x.bfloat16().float(), wt.bfloat16().float(), top.bfloat16().float()
    ref = F.conv2d(x, wt, b) + F.interpolate(top, scale_factor=2, mode="nearest")
    pc = ops().pack_conv(wt.cuda(), bias=b.cuda(), dtype=DT[dt])
    y = ops().conv2d(to_nhwc(x, DT[dt]), pc, res=to_nhwc(top, DT[dt]), res_mode=ops().RES_UP2X)
    torch.testing.assert_close(from_nhwc(y), ref, **TOL[dt])


@pytest.mark.parametrize("dt", ["f32", "bf16"])
def test_conv2d_identity_on_every_other_pixel(dt):
    """out = relu(conv1x1(mid) + identity[:, :, ::2, ::2]): conv3 of layer1's last block computed on the even pixels only (its
    output is read by layer2.0's stride-2 1x1 convs and by nothing else: resnet.py:295-315, 138-145), the epilogue reading the
    identity at (2 ho, 2 wo) — OSD_RES_DOWN2X — on every algorithm, on odd and even identity maps, with a ragged pixel tail; the
    grouped launch gives the same bits as the single ones."""
    from oneshotdet_amd import _lib
    o = ops()
    T = DT[dt]
    for (n, cin, cout, hh, ww) in ((2, 64, 256, 21, 27), (1, 64, 256, 40, 64), (3, 128, 128, 9, 10)):
        ho, wo = (hh + 1) // 2, (ww + 1) // 2
        x, wt, b, idn = rnd(n, cin, ho, wo, seed=1), rnd(cout, cin, 1, 1, seed=2) / 8, rnd(cout, seed=3), rnd(n, cout, hh, ww, seed=4)
        if dt == "bf16":
            x, wt, idn = x.bfloat16().float(), wt.bfloat16().float(), idn.bfloat16().float()
        ref = F.relu(F.conv2d(x, wt, b) + idn[:, :, ::2, ::2])
        pc = o.pack_conv(wt.cuda(), bias=b.cuda(), dtype=T)
        xx, rr = to_nhwc(x, T), to_nhwc(idn, T)
        ran = 0
        for algo in [None] + o.conv_algo_candidates(cout, False):
            try:
                y = o.conv2d(xx, pc, act=o.ACT_RELU, res=rr, res_mode=o.RES_DOWN2X, algo=algo)
            except _lib.OsdError:
                continue
            ran += 1
            torch.testing.assert_close(from_nhwc(y), ref, **TOL[dt])
        assert ran >= 4
    with pytest.raises(Exception):       # an identity map that does not cover the output
        o.conv2d(xx, pc, act=o.ACT_RELU, res=to_nhwc(rnd(3, 128, 4, 4, seed=5), T), res_mode=o.RES_DOWN2X)
    # grouped: identity maps of exactly twice the size
    sizes = [(2, 10, 12), (1, 3, 8)]
    pcs = [o.pack_conv((rnd(128, 64, 1, 1, seed=10 + i) / 8).cuda(), bias=rnd(128, seed=20 + i).cuda(), dtype=T) for i in range(2)]
    xs = [to_nhwc(rnd(n, 64, h, w, seed=30 + i), T) for i, (n, h, w) in enumerate(sizes)]
    res = [to_nhwc(rnd(n, 128, 2 * h, 2 * w, seed=40 + i), T) for i, (n, h, w) in enumerate(sizes)]
    ys = o.conv2d_multi(xs, pcs, act=o.ACT_RELU, residuals=res, res_mode=o.RES_DOWN2X, _whole=True)
    for x1, pc1, r1, y1 in zip(xs, pcs, res, ys):
        assert torch.equal(y1, o.conv2d(x1, pc1, act=o.ACT_RELU, res=r1, res_mode=o.RES_DOWN2X))
        assert torch.equal(y1, o.conv2d(x1, pc1, act=o.ACT_RELU, res=r1[:, ::2, ::2].contiguous(), res_mode=o.RES_SAME))


@pytest.mark.parametrize("dt", ["f32", "bf16"])
def test_conv2d_relu_in_and_exp_scale(dt):
    n, cin, h, w = 1, 256, 9, 7
    x, wt, b = rnd(n, cin, h, w, seed=1), rnd(4, cin, 3, 3, seed=2) / 48, rnd(4, seed=3) * 0.1
    if dt == "bf16":
        x, wt = x.bfloat16().float(), wt.bfloat16().float()
    pc = ops().pack_conv(wt.cuda(), bias=b.cuda(), dtype=DT[dt])
    y = ops().conv2d(to_nhwc(x, DT[dt]), pc, stride=2, pad=1, relu_in=True)          # P7 = conv(relu(P6)) fpn.py:98
    torch.testing.assert_close(from_nhwc(y), F.conv2d(F.relu(x), wt, b, stride=2, padding=1), **TOL[dt])
    y = ops().conv2d(to_nhwc(x, DT[dt]), pc, pad=1, act=ops().ACT_EXP_SCALE, act_scale=0.7)  # fcos.py:95-97
    torch.testing.assert_close(from_nhwc(y), torch.exp(0.7 * F.conv2d(x, wt, b, padding=1)), **TOL[dt])


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("hw", [(32, 48), (63, 63), (127, 127)])
def test_stem_conv_and_maxpool(hw, dt):
    """BaseStem.forward resnet.py:332-337 incl. odd sizes (the 127x127 query)."""
    from oneshotdet_amd import model
    h, w = hw
    x, wt = rnd(2, 3, h, w, seed=1, scale=50), rnd(64, 3, 7, 7, seed=2) / 12
    bn = [rnd(64, seed=4).abs() + 0.5, rnd(64, seed=5), rnd(64, seed=6), rnd(64, seed=7).abs() + 0.5]
    sd = {"conv1.weight": wt, "bn1.weight": bn[0], "bn1.bias": bn[1], "bn1.running_mean": bn[2], "bn1.running_var": bn[3]}
    if dt == "bf16":
        x = x.bfloat16().float()
    ref = orc.stem(x, sd, "")
    pc = ops().pack_conv(wt.cuda(), bn=[t.cuda() for t in bn], dtype=DT[dt], stem=True)
    ho, wo = ops().conv_out(h, 7, 2, 3), ops().conv_out(w, 7, 2, 3)
    hp, wp = max(2 * (ho - 1) + 7, h + 3), max(2 * (wo - 1) + 8, w + 3)
    wp += wp & 1
    xi = ops().pack_image(x.cuda(), DT[dt], hp, wp)
    y = ops().maxpool3x3s2(ops().conv2d(xi, pc, act=ops().ACT_RELU, out_hw=(ho, wo)))
    # +-150-valued pixels through a 147-tap dot product: outputs reach 1e2-1e3
    tol = dict(rtol=1e-4, atol=1e-3) if dt == "f32" else dict(rtol=5e-2, atol=0.5)
    torch.testing.assert_close(from_nhwc(y), ref, **tol)


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("shape", [(2, 256, 7, 8), (1, 256, 25, 32), (3, 256, 13, 16)])
def test_groupnorm_relu(shape, dt):
    """nn.GroupNorm(32, 256) + ReLU, fcos.py:37-38."""
    x, g, b = rnd(*shape, seed=1, scale=3) + 0.5, rnd(shape[1], seed=2).abs() + 0.5, rnd(shape[1], seed=3)
    if dt == "bf16":
        x = x.bfloat16().float()
    ref = F.relu(F.group_norm(x, 32, g, b, eps=1e-5))
    y = ops().groupnorm_relu(to_nhwc(x, DT[dt]), g.cuda(), b.cuda(), 32, 1e-5)
    torch.testing.assert_close(from_nhwc(y), ref, **(TOL[dt] if dt == "bf16" else dict(rtol=1e-4, atol=1e-4)))


@pytest.mark.parametrize("dt", ["f32", "bf16"])
def test_roi_align_reference_vectors(dt):
    """Vectors recorded through the reference's _C.roi_align_forward (tests/golden/roialign.npz)."""
    f = gu.load("roialign.npz")
    for i in range(int(f["n"])):
        scale, ph, pw, sr = f["args.%d" % i]
        x = torch.from_numpy(f["x.%d" % i])
        if dt == "bf16":
            x = x.bfloat16().float()
            ref = orc.roi_align(x, torch.from_numpy(f["rois.%d" % i]), float(scale), int(ph), int(pw), int(sr))
        else:
            ref = torch.from_numpy(f["y.%d" % i])
        y = ops().roi_align(to_nhwc(x, DT[dt]), torch.from_numpy(f["rois.%d" % i]).cuda(), float(scale), int(ph), int(pw),
                            int(sr))
        torch.testing.assert_close(y.cpu().permute(0, 3, 1, 2), ref, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("shape", [(2, 256, 7, 8), (1, 256, 100, 128), (3, 64, 5, 3)])
def test_correlate_matches_broadcast_multiply(shape, dt):
    """generalized_rcnn.py:307-311; in bf16 the product is rounded once from the fp32 product."""
    x, q = rnd(*shape, seed=1), rnd(shape[0], shape[1], seed=2)
    if dt == "bf16":
        x = x.bfloat16().float()
    ref = orc.correlate([x], [q.view(shape[0], shape[1], 1, 1)])[0]
    y = ops().correlate(to_nhwc(x, DT[dt]), q.cuda())
    if dt == "bf16":
        ref = ref.bfloat16().float()
    torch.testing.assert_close(from_nhwc(y), ref, rtol=0, atol=0)   # one fp32 multiply: bit exact


def test_shot_mean():
    x = rnd(10, 256, seed=1)
    y = ops().shot_mean(x.cuda(), 2)
    torch.testing.assert_close(y.cpu(), x.view(2, 5, 256).mean(1), rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("shots", [1, 5])
def test_query_pool_levels_equals_the_per_level_launches(dt, shots):
    """osd_query_pool_levels / _bwd (all FPN levels of the query branch in one / three launches) against osd_roialign_fwd (1 x 1) +
    osd_shot_mean and osd_shot_mean_bwd + osd_roialign_bwd + osd_cast_f32 per level: the same expressions (equal to 1e-5: the two
    compilations may contract a sample position differently by an ulp), for one and five shots, query boxes of different sizes
    (generalized_rcnn.py:20-52, 100-104, 257), and against the oracle's ROIAlign."""
    o = ops()
    T = DT[dt]
    batch, c = 3, 256
    r = batch * shots
    sizes = [(16, 16), (8, 8), (4, 5), (2, 3), (1, 1)]
    scales = [1 / 8, 1 / 16, 1 / 32, 1 / 64, 1 / 128]
    feats = [to_nhwc(rnd(r, c, h, w, seed=10 + i), T) for i, (h, w) in enumerate(sizes)]
    g = torch.Generator().manual_seed(5)
    hw = torch.rand(r, 2, generator=g) * 60 + 67            # whole-image boxes of queries between 67 and 127 pixels, as (0, 0, h, w)
    rois = torch.cat([torch.arange(r).float()[:, None], torch.zeros(r, 2), hw], 1).cuda()
    pooled = o.query_pool_levels(feats, rois, scales, batch, 2)
    for f, sc, p in zip(feats, scales, pooled):
        v = o.roi_align(f, rois, sc, 1, 1, 2)
        torch.testing.assert_close(p, o.shot_mean(v.view(r, -1), batch), rtol=1e-5, atol=1e-5)
    ref0 = orc.roi_align(from_nhwc(feats[0]).to(T).float(), rois.cpu(), scales[0], 1, 1, 2).view(batch, shots, c).mean(1)
    torch.testing.assert_close(pooled[0].cpu(), ref0, rtol=1e-5, atol=1e-5)
    dqs = [rnd(batch, c, seed=30 + i).cuda() for i in range(len(sizes))]
    outs = o.query_pool_levels_bwd(dqs, rois, [tuple(f.shape) for f in feats], scales, shots, 2, T)
    for dq, f, sc, out in zip(dqs, feats, scales, outs):
        dv = o.shot_mean_bwd(dq, shots)
        gx = o.roi_align_bwd(dv.view(-1, 1, 1, c), rois, f.shape, sc, 1, 1, 2)
        assert out.dtype == T
        torch.testing.assert_close(out.float(), o.cast_f32(gx, T).float(), rtol=1e-5 if dt == "f32" else 1e-2, atol=1e-6 if dt == "f32" else 1e-4)


def test_small_tile_convs_sharing_cus_on_two_streams_match_their_serial_results():
    """conv_sp's 128 x 128 tile asks for 76 KB of LDS, so two of its workgroups share a CU (round 6) — also workgroups of two
    different launches.  Two tower-sized 3x3 convs (each 134 workgroups) and a layer3-sized one (400) launched back and forth on two
    streams, 20 rounds: every result bit-equal to the same launch alone on the default stream."""
    o = ops()
    T = torch.bfloat16
    shapes = [(4, 40, 40, 256, 256), (4, 20, 20, 256, 256), (8, 50, 64, 256, 256)]
    xs = [to_nhwc(rnd(n, c, h, w, seed=60 + i), T) for i, (n, h, w, c, _) in enumerate(shapes)]
    pcs = [o.pack_conv((rnd(co, c, 3, 3, seed=70 + i) / 48).cuda(), bias=rnd(co, seed=80 + i).cuda(), dtype=T)
           for i, (_, _, _, c, co) in enumerate(shapes)]
    algo = 1 + 0 * 32 + 0 * 8 + 6
    want = [o.conv2d(x, pc, pad=1, act=o.ACT_RELU, algo=algo) for x, pc in zip(xs, pcs)]
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    outs = []
    for rep in range(20):
        for j, (x, pc) in enumerate(zip(xs, pcs)):
            with torch.cuda.stream(s1 if (j + rep) % 2 == 0 else s2):
                outs.append((j, o.conv2d(x, pc, pad=1, act=o.ACT_RELU, algo=algo)))
    torch.cuda.synchronize()
    for j, y in outs:
        assert torch.equal(y, want[j])


def test_empty_inputs_are_noops():
    o = ops()
    y = o.correlate(torch.zeros(0, 4, 4, 256, device="cuda"), torch.zeros(0, 256, device="cuda"))
    assert y.shape == (0, 4, 4, 256)
    r = o.roi_align(torch.zeros(1, 4, 4, 8, device="cuda"), torch.zeros(0, 5, device="cuda"), 0.5, 1, 1, 2)
    assert r.shape == (0, 1, 1, 8)


def test_sigmoid_focal_loss_fwd_bwd():
    """csrc/cuda/SigmoidFocalLoss_cuda.cu:21-101 vs the oracle's CUDA-formula restatement + autograd."""
    logits = (rnd(777, 1, seed=1) * 4).requires_grad_(True)
    targets = (torch.rand(777, generator=torch.Generator().manual_seed(2)) < 0.1).int()
    targets[5] = -1     # ignored row: contributes to neither term
    ref = orc.sigmoid_focal_loss_cuda_formula(logits, targets, 2.0, 0.25)
    gup = rnd(777, 1, seed=3)
    ref.backward(gup)
    y = ops().sigmoid_focal_loss_fwd(logits.detach().cuda(), targets.cuda(), 2.0, 0.25)
    torch.testing.assert_close(y.cpu(), ref.detach(), rtol=1e-5, atol=1e-6)
    d = ops().sigmoid_focal_loss_bwd(logits.detach().cuda(), targets.cuda(), gup.cuda(), 2.0, 0.25)
    torch.testing.assert_close(d.cpu(), logits.grad, rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("dt", ["f32", "bf16"])
def test_conv2d_every_algorithm_gives_the_same_answer(dt):
    """Every kernel generation x ring depth x tile the autotuner may pick (osd_conv_desc.algo), incl. the 8-wave
    256x256 tile, on a 3x3 conv with a ragged M tail, residual and ReLU."""
    from oneshotdet_amd import _lib
    n, cin, h, w, cout = 2, 256, 23, 19, 256
    x, wt, b, idn = rnd(n, cin, h, w, seed=1), rnd(cout, cin, 3, 3, seed=2) / 48, rnd(cout, seed=3), rnd(n, cout, h, w, seed=4)
    if dt == "bf16":
        x, wt, idn = x.bfloat16().float(), wt.bfloat16().float(), idn.bfloat16().float()
    ref = F.relu(F.conv2d(x, wt, b, padding=1) + idn)
    pc = ops().pack_conv(wt.cuda(), bias=b.cuda(), dtype=DT[dt])
    xx, rr = to_nhwc(x, DT[dt]), to_nhwc(idn, DT[dt])
    ran = 0
    # + conv_sp's general-width form forced; pixels=874: the latency-sized launches' deep-ring tiles (algos 58 / 59, round 6; bf16)
    cands = ops().conv_algo_candidates(cout, False, pixels=n * h * w) + [1 + 16 + 6]
    assert {57, 58, 59, 60} <= set(cands)
    for algo in cands:
        try:
            y = ops().conv2d(xx, pc, pad=1, act=ops().ACT_RELU, res=rr, res_mode=ops().RES_SAME, algo=algo)
        except _lib.OsdError:
            assert not (dt == "bf16" and algo in (57, 58, 59, 60))
            continue
        torch.testing.assert_close(from_nhwc(y), ref, **TOL[dt], msg=lambda m: "algo %d: %s" % (algo, m))
        ran += 1
    assert ran >= 10


@pytest.mark.parametrize("m_shape,cin,cout", [((2, 16, 24), 64, 256), ((2, 25, 32), 128, 512), ((3, 13, 19), 256, 1024), ((1, 50, 64), 256, 320),
                                                ((8, 50, 64), 256, 1024), ((4, 100, 128), 128, 512), ((2, 9, 7), 64, 64), ((1, 1, 1), 256, 128),
                                                ((1, 3, 43), 128, 192)])
def test_conv2d_pixel_stationary_pointwise_kernel_is_bit_identical(m_shape, cin, cout):
    """conv_px (algo 49, round 5: every wave keeps its 32 pixels' K values in registers, the workgroups walk (pixel block,
    64-channel chunk) units and stream only weight chunks; residual / mask / bias requested a unit ahead) against the
    one-tile-per-workgroup LDS-DMA kernel on the EXPANDING bottleneck 1x1 convs (resnet.py:295-315: conv3 + identity + ReLU) and
    conv1's data gradient (+ skip gradient + mask): same K order and MFMA roles, so BIT-identical outputs — every epilogue
    combination, K = 64 / 128 / 256, ragged pixel tails (M = 741, 126, 129, 1), units that straddle pixel blocks inside one
    workgroup, more units than workgroups and fewer, every launch repeated (race screen: a weight chunk read before its DMA
    landed, an operand set consumed before its loads returned); plus the torch reference.  Refused: cin 512, stride 2, 3x3, fp32."""
    from oneshotdet_amd import _lib
    o = ops()
    n, h, w = m_shape
    x, wt, b = rnd(n, cin, h, w, seed=1), rnd(cout, cin, 1, 1, seed=2) / cin ** 0.5, rnd(cout, seed=3)
    idn, mk = rnd(n, cout, h, w, seed=4), rnd(n, cout, h, w, seed=5)
    x, wt, idn, mk = x.bfloat16().float(), wt.bfloat16().float(), idn.bfloat16().float(), mk.bfloat16().float()
    pc = o.pack_conv(wt.cuda(), bias=b.cuda(), dtype=torch.bfloat16)
    xx, rr, mm = to_nhwc(x, torch.bfloat16), to_nhwc(idn, torch.bfloat16), to_nhwc(mk, torch.bfloat16)
    lin = F.conv2d(x, wt, b)
    cases = [(dict(), lin), (dict(act=o.ACT_RELU), F.relu(lin)),
             (dict(res=rr, res_mode=o.RES_SAME, act=o.ACT_RELU), F.relu(lin + idn)),
             (dict(mask=mm), torch.where(mk > 0, lin, torch.zeros_like(lin))),
             (dict(res=rr, res_mode=o.RES_SAME, mask=mm), torch.where(mk > 0, lin + idn, torch.zeros_like(lin)))]
    for kw, ref in cases:
        base = o.conv2d(xx, pc, algo=1 + 8 + 0, **kw)          # conv_dma, 128 x 128 tile, shallow ring
        for rep in range(6):
            algo = (o.CONV_ALGO_PX, o.CONV_ALGO_PX_WIDE)[rep & 1]      # eight waves of 16 pixels / four waves of 32
            y = o.conv2d(xx, pc, algo=algo, out=torch.full_like(base, 7.0), **kw)      # every element is written
            assert torch.equal(y, base), (sorted(kw), rep, (y.float() - base.float()).abs().max().item())
        torch.testing.assert_close(from_nhwc(y), ref, **TOL["bf16"])
    pc3 = o.pack_conv((rnd(cout, cin, 3, 3, seed=6) / 48).cuda(), bias=b.cuda(), dtype=torch.bfloat16)
    with pytest.raises(_lib.OsdError):
        o.conv2d(xx, pc3, pad=1, algo=o.CONV_ALGO_PX)
    with pytest.raises(_lib.OsdError):
        o.conv2d(xx, pc, stride=2, algo=o.CONV_ALGO_PX)
    with pytest.raises(_lib.OsdError):
        o.conv2d(to_nhwc(x, torch.float32), o.pack_conv(wt.cuda(), bias=b.cuda(), dtype=torch.float32), algo=o.CONV_ALGO_PX)
    pc512 = o.pack_conv((rnd(64, 512, 1, 1, seed=7) / 23).cuda(), bias=rnd(64, seed=8).cuda(), dtype=torch.bfloat16)
    with pytest.raises(_lib.OsdError):
        o.conv2d(to_nhwc(rnd(1, 512, 4, 4, seed=9), torch.bfloat16), pc512, algo=o.CONV_ALGO_PX)


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("cout", [2, 4])
def test_prediction_conv_data_gradient_as_a_gemm_over_the_gathered_dy(dt, cout):
    """Round 5: the data gradient of the FCOS prediction convs (cls_logits + centerness fused: 2 outputs, bbox_pred: 4; fcos.py:50-61,
    3x3 / pad 1 over the five FPN levels) as osd_pred_dy_gather -> G [pixels][64] (the nine shifted dy vectors of every pixel side by
    side) + osd_pred_dgrad_pack (W as [cin][64]) + ONE 1x1 conv with K = 64 over all levels, against autograd of F.conv2d per
    level, and the weight gradient from the same G (osd_conv2d_wgrad_pred_gathered) against autograd too.  Levels of every
    size down to 1 x 1 and 1 x 3 (all nine taps cross the border), dy stored with 4 channels."""
    o = ops()
    cin = 256
    sizes = [(2, 25, 32), (2, 13, 16), (2, 7, 8), (1, 1, 3), (3, 1, 1)]
    w = (rnd(cout, cin, 3, 3, seed=1) / 48)
    dys = [rnd(n, 4, h, ww, seed=10 + i) for i, (n, h, ww) in enumerate(sizes)]
    xs = [rnd(n, cin, h, ww, seed=20 + i) for i, (n, h, ww) in enumerate(sizes)]
    for d in dys:
        d[:, cout:] = 0.0                      # the channels past cout are stored as zeros by the loss kernel
    if dt == "bf16":
        w, dys, xs = w.bfloat16().float(), [d.bfloat16().float() for d in dys], [x.bfloat16().float() for x in xs]
    dev_dys = [to_nhwc(d, DT[dt]) for d in dys]
    dev_xs = [to_nhwc(x, DT[dt]) for x in xs]
    master = w.permute(0, 2, 3, 1).contiguous().cuda()           # [cout][3][3][cin] fp32
    g = o.pred_dy_gather(dev_dys, cout, cin)
    assert tuple(g.shape) == (sum(n * h * ww for n, h, ww in sizes), 64)
    wd = o.pred_dgrad_pack(master, cout, cin, DT[dt])
    pcd = o.PackedConv(wd, torch.zeros(cin, device="cuda"), cin, cin, cin, 64, 1, 1, cin_real=9 * cout)
    dx = o.conv2d(g.view(1, 1, g.shape[0], 64), pcd).view(-1, cin)
    dw = torch.zeros((cout, 3, 3, cin), device="cuda")
    db = torch.zeros((cout,), device="cuda")
    o.conv2d_wgrad_grouped(list(zip(dev_xs, dev_dys)), dw, 3, 3, 1, 1, cout, db=db, g=g)
    q0 = 0
    ref_dw, ref_db = torch.zeros(cout, cin, 3, 3), torch.zeros(cout)
    for (n, h, ww), d, x in zip(sizes, dys, xs):
        xr = x.clone().requires_grad_(True)
        wr = w.clone().requires_grad_(True)
        br = torch.zeros(cout, requires_grad=True)
        F.conv2d(xr, wr, br, padding=1).backward(d[:, :cout])
        got = dx[q0:q0 + n * h * ww].view(n, h, ww, cin).permute(0, 3, 1, 2).float().cpu()
        torch.testing.assert_close(got, xr.grad, **TOL[dt])
        ref_dw += wr.grad
        ref_db += br.grad
        q0 += n * h * ww
    scale = float(ref_dw.abs().max())
    assert float((dw.permute(0, 3, 1, 2).cpu() - ref_dw).abs().max()) <= (2e-2 if dt == "bf16" else 1e-4) * scale
    torch.testing.assert_close(db.cpu(), ref_db, rtol=1e-3, atol=1e-3 * float(ref_db.abs().max()))
    # the gathered matrix itself: every column of every pixel, exactly
    from oracle import launch_replay as lr
    assert torch.equal(g.float().cpu(), lr.pred_gather_launch([t.cpu() for t in dev_dys]).to(DT[dt]).float())


@pytest.mark.parametrize("dt", ["f32", "bf16"])
def test_conv2d_grouped_equals_per_level_launches(dt):
    """osd_conv2d_fwd_grouped (one launch over the FPN levels that share a tower conv, fcos.py:83-99) against one
    osd_conv2d_fwd launch per level with the same algorithm: the per-tile arithmetic is identical, so the outputs must
    match BIT FOR BIT — for every LDS-DMA algorithm, with residual + mask epilogues, ragged M tails and a 1-pixel level;
    plus the torch reference for the plain case and the per-level exp(scale_l * x) of bbox_pred."""
    from oneshotdet_amd import _lib
    o = ops()
    sizes = [(2, 25, 32), (2, 13, 16), (2, 7, 8), (2, 4, 4), (2, 1, 1)]
    cin = cout = 256
    wt, b = rnd(cout, cin, 3, 3, seed=2) / 48, rnd(cout, seed=3)
    xs = [rnd(n, cin, h, w, seed=10 + i) for i, (n, h, w) in enumerate(sizes)]
    if dt == "bf16":
        wt, xs = wt.bfloat16().float(), [x.bfloat16().float() for x in xs]
    pc = o.pack_conv(wt.cuda(), bias=b.cuda(), dtype=DT[dt])
    xx = [to_nhwc(x, DT[dt]) for x in xs]
    res = [to_nhwc(rnd(n, cout, h, w, seed=20 + i), DT[dt]) for i, (n, h, w) in enumerate(sizes)]
    msk = [to_nhwc(rnd(n, cout, h, w, seed=30 + i), DT[dt]) for i, (n, h, w) in enumerate(sizes)]
    ran = 0
    for algo in o.conv_algo_candidates(cout, False, has_mask=True) + [1 + 16 + 6]:
        try:
            ys = o.conv2d_grouped(xx, pc, pad=1, residuals=res, masks=msk, algo=algo)
        except _lib.OsdError:
            continue
        for x, r, m, y in zip(xx, res, msk, ys):
            one = o.conv2d(x, pc, pad=1, res=r, res_mode=o.RES_SAME, mask=m, algo=algo)
            assert torch.equal(one, y), "algo %d level %s" % (algo, tuple(x.shape))
        ran += 1
    assert ran >= 8
    ys = o.conv2d_grouped(xx, pc, pad=1, act=o.ACT_RELU)
    for x, y in zip(xs, ys):
        torch.testing.assert_close(from_nhwc(y), F.relu(F.conv2d(x, wt, b, padding=1)), **TOL[dt])
    # prediction conv: 4 outputs, per-level learnable Scale read from device memory (fcos.py:93-96)
    w4, b4 = rnd(4, cin, 3, 3, seed=5) / 48, rnd(4, seed=6) * 0.1
    if dt == "bf16":
        w4 = w4.bfloat16().float()
    p4 = o.pack_conv(w4.cuda(), bias=b4.cuda(), dtype=DT[dt])
    scales = torch.tensor([0.5, 1.0, 1.5, 0.25, 2.0], device="cuda")
    ys = o.conv2d_grouped(xx, p4, pad=1, act=o.ACT_EXP_SCALE, act_scale_devs=[scales[i:i + 1] for i in range(5)])
    for i, (x, y) in enumerate(zip(xs, ys)):
        ref = torch.exp(F.conv2d(x, w4, b4, padding=1) * float(scales[i]))
        torch.testing.assert_close(from_nhwc(y)[:, :4], ref, **TOL[dt])


def test_conv2d_grouped_rejects_bad_arguments():
    from oneshotdet_amd import _lib
    o = ops()
    pc = o.pack_conv((rnd(64, 64, 3, 3, seed=1) / 24).cuda(), bias=rnd(64, seed=2).cuda(), dtype=torch.float32)
    xs = [to_nhwc(rnd(1, 64, 4, 4, seed=3), torch.float32)] * 13
    with pytest.raises(_lib.OsdError):
        o.conv2d_grouped(xs, pc, pad=1, _whole=True)  # more than OSD_CONV_MAX_SEG = 12 segments
    with pytest.raises(_lib.OsdError):
        o.conv2d_grouped(xs[:2], pc, pad=1, algo=40)  # register-staged algorithms cannot run grouped


def test_conv2d_software_pipelined_row_reuse_kernel_matches_the_dma_kernel():
    """conv_igemm_sp.hip (tile id 6, variant 1) on the widths its padded-image form takes (64 / 128 / 256): 3x3 / stride 1 / pad 1,
    pixel rows fetched once per filter row into a zero-padded LDS image, fragments read one half stage ahead, the barrier in the
    middle of a stage, LDS-DMA as bounds-checked buffer loads (zero padding by the hardware).  It sums K in (r, c, s) order instead
    of the LDS-DMA kernel's (r, s, c), so it equals that kernel to fp32 rounding (bf16 outputs: at most one bf16 ulp apart, few
    elements differing) and ITSELF bit for bit over repeats (race screen: a fragment read before its DMA has landed, or a DMA into
    a buffer still being read, shows up as a changed tile).  Shapes: every width, tiles straddling images (H*W not a multiple of
    256), vertical borders inside a tile, ragged M and Cout tails, 1..4 channel slabs, all epilogues, a grouped launch; the
    128 x 128 tile (algo 7, the id of the retired row-reuse kernel) gives the same bits."""
    from oneshotdet_amd import _lib
    o = ops()
    SP, DMA = 1 + 8 + 6, 1 + 8 + 4
    for (n, h, w, cin, cout) in [(2, 50, 64, 256, 256), (3, 13, 128, 128, 256), (1, 5, 256, 64, 320), (8, 100, 128, 256, 256),
                                 (2, 7, 64, 64, 260), (1, 3, 64, 192, 256), (5, 9, 64, 256, 512)]:
        x = to_nhwc(rnd(n, cin, h, w, seed=1), torch.bfloat16)
        wt = rnd(cout, cin, 3, 3, seed=2) / (cin * 9) ** 0.5
        pc = o.pack_conv(wt.cuda(), bias=rnd(cout, seed=3).cuda(), dtype=torch.bfloat16)
        res = to_nhwc(rnd(n, pc.cout_store, h, w, seed=4), torch.bfloat16)
        mask = to_nhwc(rnd(n, pc.cout_store, h, w, seed=5), torch.bfloat16)
        for kw in (dict(), dict(res=res, res_mode=o.RES_SAME, act=o.ACT_RELU), dict(mask=mask)):
            ref = o.conv2d(x, pc, pad=1, algo=DMA, **kw)
            y = o.conv2d(x, pc, pad=1, algo=SP, **kw)
            d = (y.float() - ref.float()).abs()
            assert bool((d <= 2.0 ** -7 * ref.float().abs().clamp(min=1.0)).all()), (n, h, w, cin, cout, sorted(kw), d.max().item())
            assert (d > 0).float().mean().item() < 0.05
            for rep in range(6):
                assert torch.equal(o.conv2d(x, pc, pad=1, algo=SP, **kw), y), (n, h, w, cin, cout, sorted(kw), rep)
        assert bool((o.conv2d(x, pc, pad=1, mask=mask, algo=SP)[mask <= 0] == 0).all())
    # fp32 reference of the op itself on one shape
    n, h, w, cin, cout = 2, 20, 64, 64, 256
    xf, wf, bf = rnd(n, cin, h, w, seed=7), rnd(cout, cin, 3, 3, seed=8) / (cin * 9) ** 0.5, rnd(cout, seed=9)
    pc = o.pack_conv(wf.cuda(), bias=bf.cuda(), dtype=torch.bfloat16)
    y = o.conv2d(to_nhwc(xf, torch.bfloat16), pc, pad=1, algo=SP)
    ref = torch.nn.functional.conv2d(xf.bfloat16().float(), wf.bfloat16().float(), bf, padding=1)
    np.testing.assert_allclose(y.float().cpu().permute(0, 3, 1, 2).numpy(), ref.numpy(), rtol=2e-2, atol=2e-2)
    # grouped launch over three levels
    xs = [to_nhwc(rnd(2, 256, 24, 128, seed=11), torch.bfloat16), to_nhwc(rnd(2, 256, 12, 64, seed=12), torch.bfloat16),
          to_nhwc(rnd(1, 256, 3, 64, seed=13), torch.bfloat16)]
    wt = rnd(256, 256, 3, 3, seed=13) / (256 * 9) ** 0.5
    pc = o.pack_conv(wt.cuda(), bias=rnd(256, seed=14).cuda(), dtype=torch.bfloat16)
    for xa, ya in zip(xs, o.conv2d_grouped(xs, pc, pad=1, algo=SP, _whole=True)):
        assert torch.equal(ya, o.conv2d(xa, pc, pad=1, algo=SP))
    # algo 7 (the retired row-reuse kernel's id) now names conv_sp's 128 x 128 tile: same K order, bit-identical
    for xa, ya in zip(xs, o.conv2d_grouped(xs, pc, pad=1, algo=1 + 6, _whole=True)):
        assert torch.equal(ya, o.conv2d(xa, pc, pad=1, algo=SP))


def test_conv2d_software_pipelined_kernel_on_a_128_channel_tile():
    """conv_igemm_sp.hip with <= 128 output channels (round 5; layer2's 3x3 convs, resnet.py:295-315 with 128 bottleneck channels):
    the 256-pixel x 128-channel tile on 4 x 2 waves — two weight DMA instructions per wave and stage instead of four, the same
    per-wave 64 x 64 tile and K order as the 128-pixel form.  Against the LDS-DMA kernel to fp32 summation order (at most one bf16
    ulp, few elements differing), itself bit for bit over repeats (race screen), on the padded-image widths and the general-width
    form, tiles straddling images and the M tail, ragged channel counts, all epilogues, a grouped launch; and the fp32 reference."""
    from oneshotdet_amd import _lib
    o = ops()
    SP, GEN, HALF, DMA = 1 + 8 + 6, 1 + 16 + 6, 1 + 24 + 6, 1 + 8 + 0
    for (n, h, w, cin, cout) in [(8, 100, 128, 128, 128), (2, 50, 64, 128, 128), (3, 13, 128, 64, 96), (1, 5, 256, 256, 128),
                                 (2, 80, 104, 128, 128), (3, 23, 19, 128, 72), (8, 13, 16, 64, 128), (1, 3, 257, 192, 100)]:
        x = to_nhwc(rnd(n, cin, h, w, seed=1), torch.bfloat16)
        wt = rnd(cout, cin, 3, 3, seed=2) / (cin * 9) ** 0.5
        pc = o.pack_conv(wt.cuda(), bias=rnd(cout, seed=3).cuda(), dtype=torch.bfloat16)
        res = to_nhwc(rnd(n, pc.cout_store, h, w, seed=4), torch.bfloat16)
        mask = to_nhwc(rnd(n, pc.cout_store, h, w, seed=5), torch.bfloat16)
        for kw in (dict(), dict(res=res, res_mode=o.RES_SAME, act=o.ACT_RELU), dict(mask=mask), dict(res=res, res_mode=o.RES_SAME, mask=mask)):
            ref = o.conv2d(x, pc, pad=1, algo=DMA, **kw)
            y = o.conv2d(x, pc, pad=1, algo=SP, **kw)
            d = (y.float() - ref.float()).abs()
            assert bool((d <= 2.0 ** -7 * ref.float().abs().clamp(min=1.0)).all()), (n, h, w, cin, cout, sorted(kw), d.max().item())
            assert (d > 0).float().mean().item() < 0.05
            for rep in range(5):
                assert torch.equal(o.conv2d(x, pc, pad=1, algo=SP, **kw), y), (n, h, w, cin, cout, sorted(kw), rep)
            if w in (64, 128, 256):       # the general-width form forced on a padded-image width: the same K order, bit-identical
                assert torch.equal(o.conv2d(x, pc, pad=1, algo=GEN, **kw), y), ("general width", n, h, w, cin, cout, sorted(kw))
        with pytest.raises(_lib.OsdError):
            o.conv2d(x, pc, pad=1, algo=HALF)      # the 128-pixel tile exists for 256-channel tiles only
    n, h, w, cin, cout = 2, 20, 64, 128, 128
    xf, wf, bf = rnd(n, cin, h, w, seed=7), rnd(cout, cin, 3, 3, seed=8) / (cin * 9) ** 0.5, rnd(cout, seed=9)
    pc = o.pack_conv(wf.cuda(), bias=bf.cuda(), dtype=torch.bfloat16)
    y = o.conv2d(to_nhwc(xf, torch.bfloat16), pc, pad=1, algo=SP)
    ref = torch.nn.functional.conv2d(xf.bfloat16().float(), wf.bfloat16().float(), bf, padding=1)
    np.testing.assert_allclose(y.float().cpu().permute(0, 3, 1, 2).numpy(), ref.numpy(), rtol=2e-2, atol=2e-2)
    xs = [to_nhwc(rnd(2, 128, 24, 128, seed=11), torch.bfloat16), to_nhwc(rnd(2, 128, 12, 64, seed=12), torch.bfloat16),
          to_nhwc(rnd(1, 128, 5, 7, seed=13), torch.bfloat16)]
    for xa, ya in zip(xs, o.conv2d_grouped(xs, pc, pad=1, algo=SP, _whole=True)):
        assert torch.equal(ya, o.conv2d(xa, pc, pad=1, algo=SP))


def test_conv2d_software_pipelined_kernel_on_any_width():
    """conv_igemm_sp.hip's general-width form (GENW: consecutive pixel rows + two halo instructions in LDS, the line borders
    masked in registers).  (i) Forced on the widths the padded-image form accepts (algo 23 = tile 6, variant 2) it must be
    BIT-IDENTICAL to that form (algo 15) — same K order, and a masked fragment is the exact zero a pad row held — on every
    repeat (race screen).  (ii) Picked by algo 15 itself on every other width — BASELINE.json configs[4]'s 104 / 164 (640 x 832,
    1024 x 1312 at stride 8), a resized COCO image's 168, the tower's P5-P7 widths 32 / 16 / 8, odd and tiny widths, tiles that
    straddle lines, images and the M tail — it equals the LDS-DMA kernel to fp32 summation order (bf16 outputs: within one bf16
    ulp, few elements differing) and itself bit for bit.  (iii) One grouped launch over P3..P7 of a 640 x 832 batch equals the
    per-level launches bit for bit."""
    o = ops()
    SP, GEN, HALF, DMA, SMALL = 1 + 8 + 6, 1 + 16 + 6, 1 + 24 + 6, 1 + 8 + 4, 1 + 6
    for (n, h, w, cin, cout) in [(2, 50, 64, 256, 256), (3, 13, 128, 128, 256), (1, 5, 256, 64, 320), (2, 7, 64, 64, 260),
                                 (5, 9, 64, 256, 512)]:
        x = to_nhwc(rnd(n, cin, h, w, seed=1), torch.bfloat16)
        wt = rnd(cout, cin, 3, 3, seed=2) / (cin * 9) ** 0.5
        pc = o.pack_conv(wt.cuda(), bias=rnd(cout, seed=3).cuda(), dtype=torch.bfloat16)
        res = to_nhwc(rnd(n, pc.cout_store, h, w, seed=4), torch.bfloat16)
        mask = to_nhwc(rnd(n, pc.cout_store, h, w, seed=5), torch.bfloat16)
        for kw in (dict(), dict(res=res, res_mode=o.RES_SAME, act=o.ACT_RELU), dict(mask=mask)):
            ref = o.conv2d(x, pc, pad=1, algo=SP, **kw)
            for rep in range(4):
                y = o.conv2d(x, pc, pad=1, algo=GEN, **kw)
                assert torch.equal(y, ref), (n, h, w, cin, cout, sorted(kw), rep, (y.float() - ref.float()).abs().max().item())
            for rep in range(3):       # the 128-pixel tile (algo 31): same K order, same per-pixel arithmetic
                y = o.conv2d(x, pc, pad=1, algo=HALF, **kw)
                assert torch.equal(y, ref), ("half tile", n, h, w, cin, cout, sorted(kw), rep, (y.float() - ref.float()).abs().max().item())
                y = o.conv2d(x, pc, pad=1, algo=SMALL, **kw)      # 128 pixels x 128 channels (algo 7)
                assert torch.equal(y, ref), ("small tile", n, h, w, cin, cout, sorted(kw), rep, (y.float() - ref.float()).abs().max().item())
    for (n, h, w, cin, cout) in [(2, 80, 104, 256, 256), (1, 128, 164, 256, 256), (2, 100, 168, 64, 256), (8, 25, 32, 256, 256),
                                 (8, 13, 16, 256, 256), (8, 7, 8, 256, 256), (3, 23, 19, 128, 320), (2, 5, 3, 64, 256),
                                 (4, 4, 1, 64, 256), (1, 1, 300, 64, 256), (1, 3, 257, 192, 512)]:
        x = to_nhwc(rnd(n, cin, h, w, seed=1), torch.bfloat16)
        wt = rnd(cout, cin, 3, 3, seed=2) / (cin * 9) ** 0.5
        pc = o.pack_conv(wt.cuda(), bias=rnd(cout, seed=3).cuda(), dtype=torch.bfloat16)
        res = to_nhwc(rnd(n, pc.cout_store, h, w, seed=4), torch.bfloat16)
        mask = to_nhwc(rnd(n, pc.cout_store, h, w, seed=5), torch.bfloat16)
        for kw in (dict(), dict(res=res, res_mode=o.RES_SAME, act=o.ACT_RELU), dict(mask=mask)):
            ref = o.conv2d(x, pc, pad=1, algo=DMA, **kw)
            y = o.conv2d(x, pc, pad=1, algo=SP, **kw)
            d = (y.float() - ref.float()).abs()
            assert bool((d <= 2.0 ** -7 * ref.float().abs().clamp(min=1.0)).all()), (n, h, w, cin, cout, sorted(kw), d.max().item())
            assert (d > 0).float().mean().item() < 0.05
            for rep in range(3):
                assert torch.equal(o.conv2d(x, pc, pad=1, algo=SP, **kw), y), (n, h, w, rep)
                assert torch.equal(o.conv2d(x, pc, pad=1, algo=HALF, **kw), y), ("half tile", n, h, w, rep)
                assert torch.equal(o.conv2d(x, pc, pad=1, algo=SMALL, **kw), y), ("small tile", n, h, w, rep)
    # fp32 reference of the op itself on a width with borders inside every fragment
    n, h, w, cin, cout = 2, 11, 13, 64, 256
    xf, wf, bf = rnd(n, cin, h, w, seed=7), rnd(cout, cin, 3, 3, seed=8) / (cin * 9) ** 0.5, rnd(cout, seed=9)
    pc = o.pack_conv(wf.cuda(), bias=bf.cuda(), dtype=torch.bfloat16)
    y = o.conv2d(to_nhwc(xf, torch.bfloat16), pc, pad=1, algo=SP)
    ref = torch.nn.functional.conv2d(xf.bfloat16().float(), wf.bfloat16().float(), bf, padding=1)
    np.testing.assert_allclose(y.float().cpu().permute(0, 3, 1, 2).numpy(), ref.numpy(), rtol=2e-2, atol=2e-2)
    # P3..P7 of a 2 x 640 x 832 batch in ONE launch
    xs = [to_nhwc(rnd(2, 256, hh, ww, seed=20 + i), torch.bfloat16) for i, (hh, ww) in enumerate([(80, 104), (40, 52), (20, 26), (10, 13), (5, 7)])]
    wt = rnd(256, 256, 3, 3, seed=13) / (256 * 9) ** 0.5
    pc = o.pack_conv(wt.cuda(), bias=rnd(256, seed=14).cuda(), dtype=torch.bfloat16)
    for algo in (SP, HALF, SMALL):
        ga = o.conv2d_grouped(xs, pc, pad=1, algo=algo, _whole=True)
        for xa, ya in zip(xs, ga):
            assert torch.equal(ya, o.conv2d(xa, pc, pad=1, algo=SP))
    # still refused: 1x1, stride 2, channel counts that are not whole 64-channel slabs
    from oneshotdet_amd import _lib
    with pytest.raises(_lib.OsdError):
        o.conv2d(to_nhwc(rnd(1, 256, 8, 100, seed=1), torch.bfloat16), pc, stride=2, pad=1, algo=SP)


@pytest.mark.parametrize("dt", ["f32", "bf16"])
def test_roi_align_module_backward_matches_oracle_autograd(dt):
    """layers.ROIAlign is differentiable like the reference's (layers/roi_align.py:27-44 -> _C.roi_align_backward): the
    gradient w.r.t. the input equals autograd through the oracle's ROIAlign (tap weights are constants)."""
    from oneshotdet_amd import layers
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 16, 13, 11, generator=g)
    rois = torch.tensor([[0, 1.5, 2.0, 30.0, 40.0], [1, 0.0, 0.0, 43.0, 51.0], [1, 10.0, 5.0, 12.0, 9.0],
                         [0, -8.0, -4.0, 20.0, 18.0]], dtype=torch.float32)
    for (ph, pw, sr, scale) in ((1, 1, 2, 0.25), (7, 7, 2, 0.25), (3, 2, 0, 0.125)):
        xr = x.clone().requires_grad_(True)
        ref = orc.roi_align(xr, rois, scale, ph, pw, sr)
        w = torch.randn(ref.shape, generator=g)
        (ref * w).sum().backward()
        xd = x.to(DT[dt]).cuda().to(memory_format=torch.channels_last).requires_grad_(True)
        mod = layers.ROIAlign((ph, pw), scale, sr)
        y = mod(xd, rois.cuda())
        assert y.shape == ref.shape and y.requires_grad
        (y * w.cuda()).sum().backward()
        tol = dict(rtol=1e-4, atol=1e-5) if dt == "f32" else dict(rtol=2e-2, atol=2e-2)
        torch.testing.assert_close(y.detach().float().cpu(), ref.detach() if dt == "f32" else
                                   orc.roi_align(x.to(DT[dt]).float(), rois, scale, ph, pw, sr), **tol)
        assert xd.grad.dtype == xd.dtype and xd.grad.shape == xd.shape
        torch.testing.assert_close(xd.grad.float().cpu(), xr.grad, **tol)


def test_conv_algorithm_cache_is_keyed_on_the_full_geometry():
    """A tuned 3x3 conv on a [*, 16, 128] map (row-reuse kernel allowed: W = 128) must not hand its algorithm to the
    transposed [*, 128, 16] map, which has the same number of output pixels but a width that kernel does not support."""
    o = ops()
    w = rnd(256, 256, 3, 3, seed=1, scale=0.05)
    pc = o.pack_conv(w.cuda(), bias=torch.zeros(256).cuda(), dtype=torch.bfloat16)
    for shape in ((2, 256, 16, 128), (2, 256, 128, 16)):
        x = rnd(*shape, seed=2)
        with o.tuning():
            y = o.conv2d(to_nhwc(x, torch.bfloat16), pc, pad=1)
        ref = F.conv2d(x.bfloat16().float(), w.bfloat16().float(), padding=1)
        torch.testing.assert_close(y.float().permute(0, 3, 1, 2).cpu(), ref, rtol=2e-2, atol=2e-1)
    keys = [k for k in o.ALGO_CACHE if k[0] == o.OSD_BF16 and k[4:8] == (256, 256, 3, 3)]
    assert len({k[1:4] for k in keys}) >= 2


@pytest.mark.parametrize("dt", ["f32", "bf16"])
def test_conv2d_multi_own_weights_strides_and_topdown_add(dt):
    """osd_conv2d_fwd_multi: segments with their OWN weights / bias (both towers, or target + query backbone), stride 2,
    the same-size residual with ReLU, and the nearest-2x top-down addend (FPN lateral, fpn.py:59-64) — every output equals
    the single launch of its own (bit for bit: same kernel, same K order) ."""
    o = ops()
    T = DT[dt]
    sizes = [(2, 20, 24), (3, 6, 8), (1, 13, 10)]
    ws = [(rnd(128, 64, 1, 1, seed=10 + i) / 8).cuda() for i in range(3)]
    bs = [rnd(128, seed=20 + i).cuda() for i in range(3)]
    pcs = [o.pack_conv(w, bias=b, dtype=T) for w, b in zip(ws, bs)]
    xs = [to_nhwc(rnd(n, 64, h, w, seed=30 + i), T) for i, (n, h, w) in enumerate(sizes)]
    # 1x1 stride 2 (the first conv of a stage, STRIDE_IN_1X1), ReLU
    ys = o.conv2d_multi(xs, pcs, stride=2, act=o.ACT_RELU, _whole=True)
    for x, pc, y in zip(xs, pcs, ys):
        assert torch.equal(y, o.conv2d(x, pc, stride=2, act=o.ACT_RELU, algo=None))
    # same-size residual + ReLU (conv3 of a bottleneck)
    res = [to_nhwc(rnd(n, 128, h, w, seed=40 + i), T) for i, (n, h, w) in enumerate(sizes)]
    ys = o.conv2d_multi(xs, pcs, act=o.ACT_RELU, residuals=res, _whole=True)
    for x, pc, r, y in zip(xs, pcs, res, ys):
        assert torch.equal(y, o.conv2d(x, pc, act=o.ACT_RELU, res=r, res_mode=o.RES_SAME))
    # top-down add: addend of exactly half size (even maps only)
    ev = [(2, 20, 24), (3, 6, 8)]
    top = [to_nhwc(rnd(n, 128, h // 2, w // 2, seed=50 + i), T) for i, (n, h, w) in enumerate(ev)]
    ys = o.conv2d_multi(xs[:2], pcs[:2], residuals=top, res_mode=o.RES_UP2X, _whole=True)
    for x, pc, r, y in zip(xs[:2], pcs[:2], top, ys):
        assert torch.equal(y, o.conv2d(x, pc, res=r, res_mode=o.RES_UP2X))
    # against torch on one segment (weights are really per segment: segment 1 with segment 0's weights must differ)
    ref = F.conv2d(from_nhwc(xs[1]).to(T).float(), ws[1].cpu().to(T).float(), bs[1].cpu(), stride=2).relu()
    y1 = o.conv2d_multi(xs, pcs, stride=2, act=o.ACT_RELU, _whole=True)[1]
    torch.testing.assert_close(from_nhwc(y1), ref, **TOL[dt])
    assert not torch.equal(y1, o.conv2d(xs[1], pcs[0], stride=2, act=o.ACT_RELU))


@pytest.mark.parametrize("dt", ["f32", "bf16"])
def test_correlate_levels_forward_and_query_gradient(dt):
    """osd_correlate_levels / osd_correlate_bwd_query_levels: all FPN levels in one launch each equal the per-level calls
    (forward bit for bit; the query gradient to the order of its atomics) and the broadcast multiply / its autograd."""
    o = ops()
    T = DT[dt]
    n, c = 3, 256
    sizes = [(25, 32), (13, 16), (7, 8), (4, 4), (1, 1)]
    xs = [to_nhwc(rnd(n, c, h, w, seed=60 + i), T) for i, (h, w) in enumerate(sizes)]
    gs = [to_nhwc(rnd(n, c, h, w, seed=70 + i), T) for i, (h, w) in enumerate(sizes)]
    qs = [rnd(n, c, seed=80 + i).cuda() for i in range(len(sizes))]
    ys = o.correlate_levels(xs, qs)
    dqs = o.correlate_bwd_query_levels(gs, xs)
    for x, g, q, y, dq in zip(xs, gs, qs, ys, dqs):
        assert torch.equal(y, o.correlate(x, q))
        ref = (x.float() * q[:, None, None, :]).to(T)
        torch.testing.assert_close(y.float(), ref.float(), rtol=1e-6 if dt == "f32" else 1e-2, atol=1e-6)
        dref = (g.float() * x.float()).sum(dim=(1, 2))
        torch.testing.assert_close(dq, dref, rtol=1e-4, atol=1e-3)
        torch.testing.assert_close(dq, o.correlate_bwd_query(g, x), rtol=1e-5, atol=1e-4)


def test_conv2d_drops_a_cached_algorithm_the_library_refuses():
    """a tuner cache entry that names a kernel this library does not build for the shape (ids were reused across library
    generations; a file can be hand-edited) must not fail every step: the entry is dropped and the library's own choice runs."""
    o = ops()
    x = to_nhwc(rnd(1, 64, 9, 11, seed=5), torch.bfloat16)
    wt = rnd(64, 64, 1, 1, seed=6) / 8
    pc = o.pack_conv(wt.cuda(), bias=torch.zeros(64).cuda(), dtype=torch.bfloat16)
    ref = o.conv2d(x, pc, algo=0)
    key = (o.OSD_BF16, 1, 9, 11, 64, 64, 1, 1, 1, 0, o.RES_NONE, o.ACT_NONE, 0, False)
    o.ALGO_CACHE[key] = 41                       # the retired persistent pointwise kernel: OSD_ERR_UNSUPPORTED
    try:
        y = o.conv2d(x, pc)
        assert key not in o.ALGO_CACHE
        assert torch.equal(y, ref)
    finally:
        o.ALGO_CACHE.pop(key, None)


def test_prediction_conv_patch_kernel_matches_the_tile_kernels():
    """conv_pred.hip (algo 51, round 6): the FCOS prediction convs' forward (fcos.py:50-61, 91-97: 3x3 256 -> 2 / 4 channels, bbox with
    exp(scale_l x)) on an 8 x 32 output patch per workgroup with the input patch staged once per 64-channel slab.  Against the
    256 x 16 LDS-DMA tile (another fp32 summation order: within one bf16 rounding step, few elements differing) and the fp32
    reference; single maps and all five levels in one grouped launch (P3 .. P7 sizes, ragged maps smaller than a patch, several
    images), the learnable per-level Scale read on the device; what it does not cover is refused."""
    from oneshotdet_amd import _lib
    o = ops()
    PRED, DMA = 51, 1 + 0 * 8 + 3
    assert PRED in o.conv_algo_candidates(4, False) and PRED not in o.conv_algo_candidates(4, False, has_mask=True)
    for cout in (2, 4):
        wt = rnd(cout, 256, 3, 3, seed=2) / 48
        pc = o.pack_conv(wt.cuda(), bias=rnd(cout, seed=3).cuda() * 0.1, dtype=torch.bfloat16)
        sizes = [(2, 100, 128), (2, 50, 64), (2, 25, 32), (2, 13, 16), (2, 7, 8), (3, 9, 33), (1, 1, 1), (1, 8, 32), (1, 17, 65)]
        xs = [torch.relu(to_nhwc(rnd(n, 256, h, w, seed=10 + i), torch.bfloat16)) for i, (n, h, w) in enumerate(sizes)]
        for x in xs:
            ref = o.conv2d(x, pc, pad=1, algo=DMA)
            y = o.conv2d(x, pc, pad=1, algo=PRED)
            d = (y.float() - ref.float()).abs()
            assert bool((d <= 2.0 ** -7 * ref.float().abs().clamp(min=1.0)).all()), (cout, tuple(x.shape), d.max().item())
            assert (d > 0).float().mean().item() < 0.05
            for rep in range(3):
                assert torch.equal(o.conv2d(x, pc, pad=1, algo=PRED), y)
        # all levels in one launch, exp(scale_l * x) with the scales on the device
        scales = [torch.tensor([0.5 + 0.25 * i], device="cuda") for i in range(5)]
        ys = o.conv2d_grouped(xs[:5], pc, pad=1, act=o.ACT_EXP_SCALE, act_scale_devs=scales, algo=PRED, _whole=True)
        rs = o.conv2d_grouped(xs[:5], pc, pad=1, act=o.ACT_EXP_SCALE, act_scale_devs=scales, algo=DMA, _whole=True)
        for ya, ra in zip(ys, rs):
            d = (ya.float() - ra.float()).abs()
            assert bool((d <= 2.0 ** -6 * ra.float().abs().clamp(min=1e-3)).all()), (cout, tuple(ya.shape), d.max().item())
        # the fp32 reference
        xf = rnd(2, 256, 23, 40, seed=31)
        y = o.conv2d(to_nhwc(xf, torch.bfloat16), pc, pad=1, algo=PRED)
        ref = F.conv2d(xf.bfloat16().float(), wt.bfloat16().float(), pc.bias[:cout].cpu(), padding=1)
        np.testing.assert_allclose(y.float().cpu()[..., :cout].permute(0, 3, 1, 2).numpy(), ref.numpy(), rtol=2e-2, atol=2e-2)
    # refused: wide convs, strides, 1x1
    wide = o.pack_conv((rnd(64, 256, 3, 3, seed=4) / 48).cuda(), bias=torch.zeros(64).cuda(), dtype=torch.bfloat16)
    with pytest.raises(_lib.OsdError):
        o.conv2d(xs[2], wide, pad=1, algo=PRED)
    with pytest.raises(_lib.OsdError):
        o.conv2d(xs[2], pc, pad=1, stride=2, algo=PRED)
