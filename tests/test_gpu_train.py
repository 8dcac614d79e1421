"""GPU (-m gpu): training path — weight-gradient / GroupNorm-backward / loss kernels against PyTorch autograd on CPU,
and the whole forward+backward against the oracle's autograd gradients stored in tests/golden/train_small.npz.

Tolerances (set from tools/grad_stats.py on MI355X): gradients are compared relative to the tensor's absmax on 256 sampled
elements per tensor.  fp32: every element within 5e-4 (measured <= 2.5e-4), cosine >= 0.9995, relative L2 <= 2e-2; the one
case with a documented ReLU-mask flip (MASK_FLIP_CASES) allows 2 elements above 5e-3, none above 2e-2.  bf16: relative L2
<= 0.35 and cosine >= 0.96 per tensor (measured 0.26 / 0.9715 at worst), Scale gradients against their conditioning sum.
Losses: fp32 rtol 1e-4, bf16 rtol 3e-2."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import golden_utils as gu
from oneshotdet_amd import spec, synth
from oracle import hotpath_ref as orc

pytestmark = pytest.mark.gpu
DT = {"f32": torch.float32, "bf16": torch.bfloat16}


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).float()


def to_nhwc(x, dtype):
    return x.permute(0, 2, 3, 1).contiguous().to(dtype).cuda()


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("case", [(2, 128, 13, 17, 128, 1, 1, 0), (1, 256, 16, 12, 128, 1, 2, 0), (2, 128, 9, 11, 256, 3, 1, 1),
                                  (1, 256, 14, 10, 256, 3, 2, 1), (3, 256, 25, 32, 512, 3, 1, 1)])
def test_conv_wgrad_matches_autograd(case, dt):
    from oneshotdet_amd import ops
    n, cin, h, w, cout, k, s, p = case
    x = rnd(n, cin, h, w, seed=1)
    wt = (rnd(cout, cin, k, k, seed=2) / np.sqrt(cin * k * k)).requires_grad_(True)
    ho, wo = ops.conv_out(h, k, s, p), ops.conv_out(w, k, s, p)
    dy = rnd(n, cout, ho, wo, seed=3)
    scale = rnd(cout, seed=4).abs() + 0.5
    if dt == "bf16":
        x, dy = x.bfloat16().float(), dy.bfloat16().float()
    (F.conv2d(x, wt * scale.view(-1, 1, 1, 1), None, stride=s, padding=p) * dy).sum().backward()
    dw = torch.zeros(cout, k, k, cin, device="cuda")
    ops.conv2d_wgrad(to_nhwc(x, DT[dt]), to_nhwc(dy, DT[dt]), dw, k, k, s, p, cout, scale=scale.cuda())
    ref = wt.grad.permute(0, 2, 3, 1)
    tol = 2e-4 if dt == "f32" else 2e-2
    assert (dw.cpu() - ref).abs().max().item() <= tol * ref.abs().max().item()


@pytest.mark.parametrize("dt", ["f32", "bf16"])
def test_conv_wgrad_skinny_and_bias(dt):
    from oneshotdet_amd import ops
    n, cin, h, w, cout = 2, 256, 13, 16, 4
    x, dy = rnd(n, cin, h, w, seed=1), rnd(n, cout, h, w, seed=3)
    if dt == "bf16":
        x, dy = x.bfloat16().float(), dy.bfloat16().float()
    wt = torch.zeros(cout, cin, 3, 3, requires_grad=True)
    b = torch.zeros(cout, requires_grad=True)
    (F.conv2d(x, wt, b, padding=1) * dy).sum().backward()
    dyp = torch.zeros(n, h, w, 16, dtype=DT[dt], device="cuda")
    dyp[..., :cout] = to_nhwc(dy, DT[dt])
    dw, db = torch.zeros(cout, 3, 3, cin, device="cuda"), torch.zeros(cout, device="cuda")
    ops.conv2d_wgrad(to_nhwc(x, DT[dt]), dyp, dw, 3, 3, 1, 1, cout, db=db)
    torch.testing.assert_close(dw.cpu(), wt.grad.permute(0, 2, 3, 1), rtol=1e-3, atol=1e-3)
    torch.testing.assert_close(db.cpu(), b.grad, rtol=1e-3, atol=1e-3)


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("cout", [2, 4])
def test_prediction_conv_wgrad_levels_read_once_kernel(cout, dt):
    """The FCOS prediction convs' weight + bias gradient over several FPN levels (osd_conv2d_wgrad_grouped with Cout <= 4
    takes the read-once kernel: one pass over x, nine dy vectors per pixel) against autograd, incl. odd sizes whose last
    workgroup is ragged and a 1 x 1 level where every tap but the centre falls outside."""
    from oneshotdet_amd import ops
    cin = 256
    sizes = [(2, 25, 31), (2, 7, 8), (2, 1, 1), (2, 3, 2)]
    wt = (rnd(cout, cin, 3, 3, seed=2) / 48).requires_grad_(True)
    b = torch.zeros(cout, requires_grad=True)
    pairs, tot = [], 0
    for i, (n, h, w) in enumerate(sizes):
        x = rnd(n, cin, h, w, seed=10 + i).to(DT[dt]).float()
        dy = rnd(n, cout, h, w, seed=20 + i).to(DT[dt]).float()
        (F.conv2d(x, wt, b, padding=1) * dy).sum().backward()
        dyp = torch.zeros(n, h, w, 64, dtype=DT[dt])          # the data-gradient conv's K padding: 64-channel rows
        dyp[..., :cout] = dy.permute(0, 2, 3, 1)
        pairs.append((to_nhwc(x, DT[dt]), dyp.cuda()))
    dw, db = torch.zeros(cout, 3, 3, cin, device="cuda"), torch.zeros(cout, device="cuda")
    ops.conv2d_wgrad_grouped(pairs, dw, 3, 3, 1, 1, cout, db=db)
    ref = wt.grad.permute(0, 2, 3, 1)
    assert (dw.cpu() - ref).abs().max().item() <= 1e-4 * ref.abs().max().item()
    assert (db.cpu() - b.grad).abs().max().item() <= 1e-4 * b.grad.abs().max().item()


@pytest.mark.parametrize("dt", ["f32", "bf16"])
def test_groupnorm_relu_backward(dt):
    from oneshotdet_amd import ops
    x = (rnd(2, 256, 13, 16, seed=1, scale=2) + 0.3)
    dy = rnd(2, 256, 13, 16, seed=2)
    if dt == "bf16":
        x, dy = x.bfloat16().float(), dy.bfloat16().float()
    x.requires_grad_(True)
    g = (rnd(256, seed=3).abs() + 0.5).requires_grad_(True)
    b = rnd(256, seed=4).requires_grad_(True)
    (F.relu(F.group_norm(x, 32, g, b, eps=1e-5)) * dy).sum().backward()
    xx = to_nhwc(x.detach(), DT[dt])
    _, ab = ops.groupnorm_relu_train(xx, g.detach().cuda(), b.detach().cuda(), 32, 1e-5)
    dg, db = torch.zeros(256, device="cuda"), torch.zeros(256, device="cuda")
    du = ops.groupnorm_relu_bwd(xx, to_nhwc(dy, DT[dt]), ab, g.detach().cuda(), b.detach().cuda(), dg, db, 32)
    tol = dict(rtol=2e-3, atol=2e-3) if dt == "f32" else dict(rtol=5e-2, atol=5e-2)
    torch.testing.assert_close(du.float().cpu().permute(0, 3, 1, 2), x.grad, **tol)
    torch.testing.assert_close(dg.cpu(), g.grad, rtol=tol["rtol"], atol=tol["atol"] * 20)
    torch.testing.assert_close(db.cpu(), b.grad, rtol=tol["rtol"], atol=tol["atol"] * 20)


def _engine_and_inputs(dt, name="small"):
    from oneshotdet_amd import train
    B, H, W, S, qh, qw = gu.CASES[name]
    img, q = gu.case_inputs(name)
    gts = synth.make_gt_boxes(B, H, W, seed=3, max_boxes=3)
    eng = train.TrainEngine(synth.make_state_dict(spec.hot_path_shapes()), dtype=DT[dt])
    G = max(len(g) for g in gts)
    gtb = torch.zeros(B, G, 4)
    for i, g in enumerate(gts):
        gtb[i, :len(g)] = torch.from_numpy(g)
    cnt = torch.tensor([len(g) for g in gts], dtype=torch.int32)
    return eng, torch.from_numpy(img).cuda(), torch.from_numpy(q).cuda(), gtb.cuda(), cnt.cuda()


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("name", ["small", "nonsquare", "shots5", "tall", "config1", "config1x2", "ms640", "ms1024"])
def test_forward_backward_matches_oracle_autograd(name, dt):
    """Losses (R12) and parameter gradients of the whole training forward+backward vs the oracle's autograd (stored in
    train_<case>.npz; the oracle itself is pinned bit-exact to the reference's losses and gradients there).  `config1` is
    the BASELINE.json geometry (800x1024 target, 127x127 query): one image of the bs=8 benchmark step at full size.
    fp32 errors there: 1e-5 of the largest entry on most tensors, up to 1.6e-2 on the cls tower — traced
    (tools/tower_trace.py) to ONE GroupNorm output per level within 1e-8 of zero whose ReLU mask the fp32 rounding of
    x*scale+shift decides differently here and in ATen; the reference is discontinuous at that element."""
    f = gu.load("train_%s.npz" % name)
    eng, img, q, gtb, cnt = _engine_and_inputs(dt, name)
    losses = eng.forward_backward(img, q, gtb, cnt).cpu().numpy()
    assert int(losses[3]) == int(f["num_pos"])
    np.testing.assert_allclose(losses[:3], f["losses_cuda_formula"], rtol=1e-4 if dt == "f32" else 3e-2)
    _check_grads_against_fixture(eng.named_grads(), f, dt, case=name)


# fp32 cases in which ONE GroupNorm output per level lies within ~1e-8 of zero, so the fp32 rounding of x*scale+shift
# decides its ReLU mask differently here and in ATen (tools/tower_trace.py prints the element): the reference is
# discontinuous there.  The flipped element changes one row of the cls tower's weight gradient by 1-2 % of the tensor's
# largest entry and everything upstream of it by ~1e-3 (tools/grad_stats.py: config1 cls_tower.0 errors 1.6e-2, 9.4e-4,
# 5.6e-4, 1.3e-4, ...; backbone tensors up to 2.9e-3).  Every other case has no such element: max error <= 2.5e-4.
MASK_FLIP_CASES = {"config1": dict(tier1=5e-3, outliers=2),
                   # the multi-scale shapes: ONE sampled element per tensor may leave the 5e-4 band (measured: one at 1.1e-3 of the
                   # absmax in backbone.body.layer4.2.conv3.weight at 640x832, one at 5.4e-4 in layer2.0.conv1.weight at 1024x1312;
                   # every other sampled element <= 4.3e-4), none may exceed 2e-2
                   "ms640": dict(tier1=5e-4, outliers=1), "ms1024": dict(tier1=5e-4, outliers=1)}


def _check_grads_against_fixture(grads, f, dt, case=None):
    """Two-tier element bound + whole-tensor bounds on the 256 sampled elements of every fixture tensor.
    fp32: every element within 5e-4 x absmax (measured <= 2.5e-4), except in MASK_FLIP_CASES where at most `outliers`
    elements may exceed `tier1` x absmax and none 2e-2; cosine >= 0.9995 and relative L2 <= 2e-2 in all cases.
    bf16 (activations and gradients rounded to 8 bits at every layer): relative L2 <= 0.35 and cosine >= 0.96 per tensor
    (measured over the six cases: L2 <= 0.26, cosine >= 0.9715); the Scale gradients are checked by
    test_bf16_scale_gradients_within_their_conditioning."""
    checked = 0
    for key in f.files:
        if key.startswith("fullgrad_oracle.") and key.endswith(".samples"):
            k = key[len("fullgrad_oracle."):-len(".samples")]
            if dt == "bf16" and k.endswith(".scale"):
                continue
            g = grads[k].float().cpu().numpy().reshape(-1)
            idx = gu.sample_indices(g.size, "grad." + k)[:256]
            scale = float(f["fullgrad_oracle.%s.absmax" % k])
            ref = f[key]
            if scale == 0.0:
                assert np.abs(g[idx]).max() == 0.0, k
                continue
            err = np.abs(g[idx] - ref) / scale
            l2 = np.linalg.norm(g[idx] - ref) / max(np.linalg.norm(ref), 1e-30)
            cos = float(np.dot(g[idx], ref) / max(np.linalg.norm(g[idx]) * np.linalg.norm(ref), 1e-30))
            if dt == "bf16":
                assert l2 <= 0.35 and cos >= 0.96, (k, l2, cos)
            else:
                flip = MASK_FLIP_CASES.get(case)
                tier1 = flip["tier1"] if flip else 5e-4
                n_out = int((err > tier1).sum())
                assert n_out <= (flip["outliers"] if flip else 0), (k, n_out, np.sort(err)[::-1][:4])
                assert err.max() <= 2e-2, (k, err.max())
                assert l2 <= 2e-2 and cos >= 0.9995, (k, l2, cos)
            checked += 1
    assert checked >= 14


@pytest.mark.parametrize("name", ["small", "nonsquare", "shots5", "tall"])
def test_bf16_scale_gradients_within_their_conditioning(name):
    """d loss / d scale_l = sum over locations of signed terms t_p = (d loss / d conv_p) * x_p / scale_l that nearly cancel,
    so under bf16 activations its RELATIVE error is unbounded; what bf16 bounds is the error against sum |t_p|: every term
    carries the forward's relative error of the regressed distances (test_bf16_forward_close_to_reference_golden: rtol 0.2
    after ~60 layers and an exp) and these errors are systematic, not random.  The fp32 engine first validates the formula
    (its own Scale gradient = sum t_p to 1e-3, = the reference's to 1e-4 of sum |t_p|), then the bf16 engine's Scale
    gradients must lie within 0.25 * sum |t_p| of the reference's (measured: 0.084 on `small`, whose relative error is 91 %)."""
    f = gu.load("train_%s.npz" % name)
    for dt in ("f32", "bf16"):
        eng, img, q, gtb, cnt = _engine_and_inputs(dt, name)
        eng.forward_backward(img, q, gtb, cnt)
        torch.cuda.synchronize()
        scales, gscales = eng.extra["rpn.head.scales"]
        for lvl in (0, 4):
            key = "fullgrad_oracle.rpn.head.scales.%d.scale.samples" % lvl
            if key not in f.files:
                continue
            ref = float(f[key][0])
            reg = eng.last_head_out[lvl][1].float()[..., :4]                   # exp(scale * x)
            x = reg.log() / scales[lvl]
            d = eng.last_pred_grads[lvl][1].float()[..., :4]
            terms = d * x / scales[lvl]
            got = float(gscales[lvl])
            if dt == "f32":
                assert abs(float(terms.sum()) - got) <= 1e-3 * float(terms.abs().sum()) + 1e-7, (lvl, float(terms.sum()), got)
                assert abs(got - ref) <= 1e-4 * float(terms.abs().sum()) + 1e-7, (lvl, got, ref)
            else:
                assert abs(got - ref) <= 0.25 * float(terms.abs().sum()) + 1e-7, (lvl, got, ref, float(terms.abs().sum()))


def test_sgd_steps_reduce_the_loss():
    eng, img, q, gtb, cnt = _engine_and_inputs("bf16")
    w0 = eng.flat_w.clone()
    first = eng.train_step(img, q, gtb, cnt)[:3].sum().item()
    for _ in range(5):
        last = eng.train_step(img, q, gtb, cnt)[:3].sum().item()
    assert not torch.equal(w0, eng.flat_w)
    assert np.isfinite(last) and last < first


def test_forward_backward_five_shots_vs_oracle_autograd():
    """BASELINE.json configs[4] structure (S = 5 queries per image, mean-pooled: generalized_rcnn.py:100-104): the query
    branch gradient flows through shot-mean backward and ROIAlign backward.  Oracle autograd computed here on the CPU."""
    from oneshotdet_amd import train
    name = "shots5"
    B, H, W, S, qh, qw = gu.CASES[name]
    img, q = gu.case_inputs(name)
    gts = synth.make_gt_boxes(B, H, W, seed=5, max_boxes=3)
    np_sd = synth.make_state_dict(spec.hot_path_shapes())
    sd = {k: v.clone().requires_grad_(not spec.is_frozen(k)) for k, v in orc.to_torch_state_dict(np_sd).items()}
    o = orc.hot_path_forward(torch.from_numpy(img), torch.from_numpy(q), sd, shots=S)
    c, r, t, info = orc.fcos_loss(o["logits"], o["bbox_reg"], o["centerness"], gts, focal="cuda")
    (c + r + t).backward()
    eng = train.TrainEngine(np_sd, dtype=torch.float32)
    G = max(len(g) for g in gts)
    gtb = torch.zeros(B, G, 4)
    for i, g in enumerate(gts):
        gtb[i, :len(g)] = torch.from_numpy(g)
    cnt = torch.tensor([len(g) for g in gts], dtype=torch.int32)
    losses = eng.forward_backward(torch.from_numpy(img).cuda(), torch.from_numpy(q).cuda(), gtb.cuda(), cnt.cuda()).cpu()
    assert int(losses[3]) == info["num_pos"]
    np.testing.assert_allclose(losses[:3].numpy(), [c.item(), r.item(), t.item()], rtol=1e-4)
    grads = eng.named_grads()
    for k in ("supp_backbone.body.layer2.0.conv1.weight", "supp_backbone.body.layer4.2.conv3.weight",
              "supp_backbone.fpn.fpn_inner2.weight", "supp_backbone.fpn.top_blocks.p7.weight",
              "backbone.body.layer3.2.conv2.weight", "rpn.head.cls_tower.0.weight"):
        ref = sd[k].grad
        err = (grads[k].float().cpu() - ref).abs().max().item()
        assert err <= 2e-2 * ref.abs().max().item(), (k, err, ref.abs().max().item())


def test_fused_sgd_matches_torch_optim():
    """osd_sgd_momentum_multi vs torch.optim.SGD with the reference's parameter groups (solver/build.py:8-26) over three
    steps on identical gradients."""
    from oneshotdet_amd import train
    np_sd = synth.make_state_dict(spec.hot_path_shapes())
    a = train.TrainEngine(np_sd, dtype=torch.bfloat16, optimizer="fused")
    b = train.TrainEngine(np_sd, dtype=torch.bfloat16, optimizer="torch")
    g = torch.Generator(device="cuda").manual_seed(0)
    for _ in range(3):
        grad = torch.randn(a.flat_g.shape, device="cuda", generator=g) * 0.01
        a.flat_g.copy_(grad)
        b.flat_g.copy_(grad)
        a.optimizer_step()
        b.optimizer_step()
    torch.testing.assert_close(a.flat_w, b.flat_w, rtol=1e-6, atol=1e-7)
    # the fused update also wrote the forward-form packed weights (osd_sgd_momentum_pack_multi): they must be what a repack of
    # the updated masters writes, bit for bit, padding included, and every bucket must have taken the fused launch
    assert all(t["fused"] for t in a._sgd["tables"].values())
    got = [a._pack[f]["flat"].clone() for f in (0, 1)]
    a.repack()
    for f in (0, 1):
        assert torch.equal(got[f].view(torch.int16), a._pack[f]["flat"].view(torch.int16)), "packed form %d" % f
    for c in a.convs.values():
        if c.trainable and c.has_bias and c.cout % 16 != 0:
            assert torch.equal(c.pc.bias[:c.cout], c.b)


def test_sgd_step_uses_the_reference_parameter_groups():
    """One optimiser step vs torch.optim.SGD built the way solver/build.py:8-26 builds it — one group per parameter, keyed
    on the REFERENCE key name: "bias" in the key -> lr x 2 and weight decay 0, everything else (conv and GroupNorm weights
    and the Scale scalars rpn.head.scales.N.scale) BASE_LR and WEIGHT_DECAY.  Built here from state_dict() names, not from
    the engine's own lists."""
    from oneshotdet_amd import train
    np_sd = synth.make_state_dict(spec.hot_path_shapes())
    lr, wd, mom = 0.05, 0.01, 0.9
    eng = train.TrainEngine(np_sd, dtype=torch.bfloat16, lr=lr, weight_decay=wd, momentum=mom)
    g = torch.Generator(device="cuda").manual_seed(3)
    sd0 = {k: v.clone() for k, v in eng.state_dict().items() if not spec.is_frozen(k)}
    params = {k: torch.nn.Parameter(v.clone()) for k, v in sd0.items()}
    groups = [{"params": [p], "lr": lr * (2 if "bias" in k else 1), "weight_decay": 0.0 if "bias" in k else wd}
              for k, p in params.items()]
    ref = torch.optim.SGD(groups, lr, momentum=mom)
    for _ in range(2):
        eng.flat_g.copy_(torch.randn(eng.flat_g.shape, device="cuda", generator=g) * 0.01)
        for k, gr in eng.named_grads().items():
            params[k].grad = gr.detach().clone().reshape(params[k].shape)
        eng.optimizer_step()
        ref.step()
    sd1 = eng.state_dict()
    scales = [k for k in params if ".scales." in k]
    assert len(scales) == 5
    for k, p in params.items():
        torch.testing.assert_close(sd1[k], p.detach(), rtol=1e-6, atol=1e-7, msg=lambda m, k=k: "%s: %s" % (k, m))
    # the Scale scalars decay (weights group): with a zero gradient they shrink by lr * wd
    assert all((sd1[k] != sd0[k]).all() for k in scales)


def test_optimizer_state_survives_a_checkpoint(tmp_path):
    """utils/checkpoint.py:33-50 stores model + optimizer + iteration: a run resumed from save_training_checkpoint continues
    with its momentum (weights after one more step are identical to the uninterrupted run's)."""
    from oneshotdet_amd import checkpoint, train
    eng, img, q, gtb, cnt = _engine_and_inputs("f32")
    for _ in range(2):
        eng.train_step(img, q, gtb, cnt)
    path = checkpoint.save_training_checkpoint(str(tmp_path / "model_0000002.pth"), eng, 2)
    st = eng.optimizer_state_dict()
    assert st["steps"] == 2 and any(v.abs().max() > 0 for v in st["momentum_buffer"].values())
    assert set(st["momentum_buffer"]) == {k for k in eng.state_dict() if not spec.is_frozen(k)}
    eng2, it = checkpoint.resume_training(path, lambda sd: train.TrainEngine(sd, dtype=torch.float32))
    assert it == 2 and eng2._sgd["steps"] == 2
    torch.testing.assert_close(eng2._sgd["buf"], eng._sgd["buf"], rtol=0, atol=0)
    eng.train_step(img, q, gtb, cnt)
    eng2.train_step(img, q, gtb, cnt)
    eng.join(), eng2.join()
    torch.cuda.synchronize()
    a, b = eng.state_dict(), eng2.state_dict()
    # same kernels, same inputs; the weight-gradient atomics make the two runs differ in the last bits only
    for k in a:
        torch.testing.assert_close(a[k], b[k], rtol=1e-4, atol=1e-6, msg=lambda m, k=k: "%s: %s" % (k, m))
    fresh = train.TrainEngine(a, dtype=torch.float32)          # without the momentum the next step differs visibly
    assert fresh._sgd["steps"] == 0


def test_state_dict_round_trip_and_update():
    """TrainEngine.state_dict() returns the reference's names/shapes (tests/golden/state_dict_keys.json); untouched it
    equals the input bit for bit; after an SGD step exactly the trainable tensors have moved."""
    import json
    import os
    from oneshotdet_amd import train
    np_sd = synth.make_state_dict(spec.hot_path_shapes())
    eng = train.TrainEngine(np_sd, dtype=torch.float32, lr=0.05)      # large enough that every update exceeds an fp32 ulp
    sd0 = eng.state_dict()
    keys = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "state_dict_keys.json")))["shapes"]
    for k, shape in spec.hot_path_shapes().items():
        assert tuple(keys[k]) == tuple(sd0[k].shape) == tuple(shape), k
        assert np.array_equal(sd0[k].cpu().numpy(), np_sd[k]), k
    _, img, q, gtb, cnt = _engine_and_inputs("f32")
    eng.train_step(img, q, gtb, cnt)
    sd1 = eng.state_dict()
    moved = {k for k in sd0 if not torch.equal(sd0[k], sd1[k])}
    assert moved and all(not spec.is_frozen(k) for k in moved)
    # weights carry weight decay, so they move even where the gradient is exactly zero (biases / Scales may not)
    assert {k for k in sd0 if not spec.is_frozen(k) and k.endswith(".weight")} - moved == set()


def test_gradient_buckets_cover_the_flat_buffer_and_overlapped_exchange_runs():
    """dist_utils.bucket_ranges on the real parameter plan: 7 contiguous buckets covering the flat gradient buffer once,
    in buffer order.  Then the overlapped exchange itself on ONE rank over RCCL (the hooks in backward, the events, the
    communication stream, finish()): averaging over one rank must leave losses and gradients bit-identical."""
    import os
    import torch.distributed as dist
    from oneshotdet_amd import train
    from oneshotdet_amd.dist_utils import GradExchange, bucket_ranges
    eng, img, q, gtb, cnt = _engine_and_inputs("bf16")
    ranges = bucket_ranges(eng._plan, eng.flat_g.numel())
    assert [n for n, _, _ in ranges] == ["backbone.layer2", "backbone.layer3", "backbone.layer4+fpn", "supp_backbone.layer2",
                                         "supp_backbone.layer3", "supp_backbone.layer4+fpn", "head"]
    assert ranges[0][1] == 0 and ranges[-1][2] == eng.flat_g.numel()
    assert all(a[2] == b[1] for a, b in zip(ranges, ranges[1:]))
    for name, c in eng.convs.items():
        if c.trainable:
            off = (c.gw.data_ptr() - eng.flat_g.data_ptr()) // 4
            b = [n for n, lo, hi in ranges if lo <= off < hi][0]
            assert b == ("head" if name.startswith("rpn.") else name.split(".")[0] + "." + (
                "layer4+fpn" if (".layer4." in name or ".fpn." in name) else name.split(".")[2])), (name, b)
    assert not eng.exchange.active
    ref_losses = eng.forward_backward(img, q, gtb, cnt).clone()
    ref_grads = eng.flat_g.clone()
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        eng.exchange = GradExchange(eng.flat_g, ranges, None, single_rank_too=True)
        assert eng.exchange.active and eng.exchange.avg
        for _ in range(2):                                        # second step: finish() re-armed the buckets
            losses = eng.forward_backward(img, q, gtb, cnt)
            assert eng.exchange.pending == set()                  # every bucket was announced during backward
            eng.reduce_gradients()
            torch.cuda.synchronize()
            assert torch.equal(losses, ref_losses)
            torch.testing.assert_close(eng.flat_g, ref_grads, rtol=1e-3, atol=1e-5)   # atomics: order-dependent last bits
        # full steps over RCCL with the tail left un-joined (bench.py's mode): exchange, update and repack of every bucket
        # ride the communication stream, the next step joins on the events; weights stay finite and keep moving
        eng.defer_join = True
        w0 = eng.flat_w.clone()
        for _ in range(3):
            out = eng.train_step(img, q, gtb, cnt)
        assert eng._deferred is not None
        eng.join()
        torch.cuda.synchronize()
        assert torch.isfinite(out).all() and torch.isfinite(eng.flat_w).all() and not torch.equal(eng.flat_w, w0)
        assert eng._sgd["steps"] == 3
        # bf16 buckets on the wire (GradExchange(wire_dtype=bfloat16) -> osd_grad_wire_cast): with one rank the average of a
        # bucket is the bucket itself rounded once to bf16, element for element, and the fp32 buffer receives exactly that
        eng.defer_join = False
        eng.exchange = GradExchange(eng.flat_g, ranges, None, single_rank_too=True, wire_dtype=torch.bfloat16)
        eng.forward_backward(img, q, gtb, cnt)
        eng.join()
        before = None
        # the exchange has already rewritten the buffer behind backward; recompute the un-exchanged gradients to compare
        eng.reduce_gradients()
        torch.cuda.synchronize()
        got = eng.flat_g.clone()
        eng.exchange = GradExchange(eng.flat_g, ranges, None, single_rank_too=False)
        eng.forward_backward(img, q, gtb, cnt)
        torch.cuda.synchronize()
        want = eng.flat_g.bfloat16().float()
        assert torch.equal(got.bfloat16().float(), got), "the exchanged gradients are not bf16 values"
        rel = float((got - want).norm() / want.norm())
        assert rel < 5e-3, rel          # two runs of the atomically accumulated gradients differ in their last fp32 bits -> rare bf16 flips
    finally:
        dist.destroy_process_group()


def test_train_step_with_updates_behind_backward_equals_the_sequential_step():
    """train_step applies each bucket's SGD update + repack on a side stream as soon as the bucket's gradients are
    final (while the rest of backward still runs).  Three steps of it against three steps of the sequential
    forward_backward -> reduce_gradients -> optimizer_step on an identical engine: same masters, same packed weights,
    same losses.  fp32 engines: in bf16 a single rounding flip of a repacked weight makes two runs of the SAME code
    differ by 1 % after three steps at this learning rate, which would hide a real ordering bug."""
    from oneshotdet_amd import train
    a, img, q, gtb, cnt = _engine_and_inputs("f32")
    b = train.TrainEngine(synth.make_state_dict(spec.hot_path_shapes()), dtype=torch.float32)
    w0 = a.flat_w.clone()
    a.lr = b.lr = 0.002
    for _ in range(3):
        la = a.train_step(img, q, gtb, cnt)
        lb = b.forward_backward(img, q, gtb, cnt)
        b.reduce_gradients()
        b.optimizer_step()
        torch.cuda.synchronize()
        assert a._updated == set() and a._sgd["steps"] == b._sgd["steps"]
        torch.testing.assert_close(la, lb, rtol=1e-3, atol=1e-6)
    da, db = a.flat_w - w0, b.flat_w - w0
    assert float(db.abs().max()) > 1e-3                                   # the steps did move the weights
    assert float((da - db).norm() / db.norm()) < 2e-3                     # atomic-order noise only
    for form in (0, 1):
        torch.testing.assert_close(a._pack[form]["flat"], b._pack[form]["flat"], rtol=1e-3, atol=1e-4)


@pytest.mark.parametrize("case", [(3, 256, 25, 32, 512, 3, 1, 1), (2, 320, 14, 10, 384, 1, 1, 0), (1, 512, 9, 7, 256, 3, 2, 1)])
def test_conv_wgrad_every_algorithm_matches_autograd(case):
    """Every (stage shape / tile, split target) choice of the weight-gradient tuner, incl. the 256 x 256 8-wave tile
    (variant 4) with ragged channel tails (320 / 384 channels), a stride-2 conv, the fused bias gradient and the grouped
    (several FPN levels) form."""
    from oneshotdet_amd import ops
    n, cin, h, w, cout, k, s, p = case
    x = rnd(n, cin, h, w, seed=1).bfloat16().float()
    wt = (rnd(cout, cin, k, k, seed=2) / np.sqrt(cin * k * k)).requires_grad_(True)
    b = torch.zeros(cout, requires_grad=True)
    ho, wo = ops.conv_out(h, k, s, p), ops.conv_out(w, k, s, p)
    dy = rnd(n, cout, ho, wo, seed=3).bfloat16().float()
    (F.conv2d(x, wt, b, stride=s, padding=p) * dy).sum().backward()
    ref, refb = wt.grad.permute(0, 2, 3, 1), b.grad
    xx, dd = to_nhwc(x, torch.bfloat16), to_nhwc(dy, torch.bfloat16)
    cands = ops.wgrad_algo_candidates(ops.OSD_BF16, cout, cin)
    assert {(a - 1) & 15 for a in cands} >= set(range(16)) - {12}
    for algo in cands:
        dw, db = torch.zeros(cout, k, k, cin, device="cuda"), torch.zeros(cout, device="cuda")
        ops.conv2d_wgrad(xx, dd, dw, k, k, s, p, cout, db=db, algo=algo)
        assert (dw.cpu() - ref).abs().max().item() <= 2e-2 * ref.abs().max().item(), algo
        assert (db.cpu() - refb).abs().max().item() <= 2e-2 * refb.abs().max().item(), algo
    if s == 1:      # grouped: the same tensors as two "levels" -> twice the gradient
        for algo in (1 + 4, 1 + 4 + 16 * 4, 1 + 0, 1 + 5, 1 + 6 + 16 * 2, 1 + 7, 1 + 8, 1 + 10 + 16, 1 + 11, 1 + 3 + 16, 1 + 13 + 16 * 3, 1 + 14 + 16, 1 + 15):
            dw = torch.zeros(cout, k, k, cin, device="cuda")
            ops.conv2d_wgrad_grouped([(xx, dd), (xx, dd)], dw, k, k, 1, p, cout, algo=algo)
            assert (dw.cpu() - 2 * ref).abs().max().item() <= 2e-2 * 2 * ref.abs().max().item(), algo


@pytest.mark.parametrize("variant", [13, 3])
@pytest.mark.parametrize("case", [(3, 256, 25, 32, 512, 3, 1), (2, 320, 13, 17, 384, 3, 1), (2, 256, 7, 8, 256, 3, 1), (1, 264, 5, 61, 256, 1, 1),
                                  (8, 256, 50, 64, 256, 3, 1), (1, 256, 2, 19, 256, 3, 1), (5, 256, 3, 4, 272, 3, 1), (2, 256, 1, 16, 256, 3, 1),
                                  (2, 256, 27, 19, 256, 3, 2), (3, 512, 14, 10, 256, 1, 2)])
def test_conv_wgrad_pipelined_variant_matches_autograd(case, variant):
    """Variants 13 / 3 (conv_wgrad_sk.hip: 256 x 256 tile on eight waves, fragments of k step u + 1 read under the MFMAs of
    step u, buffer DMA with computed bounds; 13 with the launcher's pixel splits, 3 in team mode — teams of (tiles x taps)
    workgroups walk equal shares of the concatenated pixel axis and add a tile where the gradient changes): every split
    target / round count, ragged channel tails (320 / 384 / 264 / 272), map
    widths that are not powers of two (17, 61, 19: row and image index of a pixel come from multiply-high reciprocals), maps
    smaller than one 64-pixel stage (7 x 8, 2 x 19, 3 x 4), 1x1 convs, stride 2 (3x3 pad 1 and the 1x1 downsample), the fused
    bias gradient, the scale (folded FrozenBN) form, and several levels in one launch — against autograd on the bf16-rounded
    operands; a one-row output map is refused (OSD_ERR_UNSUPPORTED)."""
    from oneshotdet_amd import ops
    n, cin, h, w, cout, k, st = case
    p = k // 2
    ho, wo = ops.conv_out(h, k, st, p), ops.conv_out(w, k, st, p)
    if ho < 2:
        xx, dd = torch.zeros(n, h, w, cin, device="cuda", dtype=torch.bfloat16), torch.zeros(n, ho, wo, cout, device="cuda", dtype=torch.bfloat16)
        with pytest.raises(Exception, match="pipelined variants"):
            ops.conv2d_wgrad(xx, dd, torch.zeros(cout, k, k, cin, device="cuda"), k, k, st, p, cout, algo=1 + variant)
        return
    x = rnd(n, cin, h, w, seed=1).bfloat16().float()
    wt = (rnd(cout, cin, k, k, seed=2) / np.sqrt(cin * k * k)).requires_grad_(True)
    b = torch.zeros(cout, requires_grad=True)
    dy = rnd(n, cout, ho, wo, seed=3).bfloat16().float()
    (F.conv2d(x, wt, b, stride=st, padding=p) * dy).sum().backward()
    ref, refb = wt.grad.permute(0, 2, 3, 1), b.grad
    xx, dd = to_nhwc(x, torch.bfloat16), to_nhwc(dy, torch.bfloat16)
    for t in range(8 if variant == 13 else 4):
        dw, db = torch.zeros(cout, k, k, cin, device="cuda"), torch.zeros(cout, device="cuda")
        ops.conv2d_wgrad(xx, dd, dw, k, k, st, p, cout, db=db, algo=1 + variant + 16 * t)
        assert (dw.cpu() - ref).abs().max().item() <= 2e-2 * ref.abs().max().item(), t
        assert (db.cpu() - refb).abs().max().item() <= 2e-2 * refb.abs().max().item(), t
    scale = (rnd(cout, seed=4).abs() + 0.5).cuda()
    dw = torch.zeros(cout, k, k, cin, device="cuda")
    ops.conv2d_wgrad(xx, dd, dw, k, k, st, p, cout, scale=scale, algo=1 + variant)
    refs = ref * scale.cpu().view(-1, 1, 1, 1)
    assert (dw.cpu() - refs).abs().max().item() <= 2e-2 * refs.abs().max().item()
    if st != 1:
        return
    # two levels of different sizes into one gradient
    h2, w2 = max(2, h // 2), max(2, w // 2)
    x2 = rnd(n, cin, h2, w2, seed=5).bfloat16().float()
    dy2 = rnd(n, cout, h2, w2, seed=6).bfloat16().float()
    wt2 = wt.detach().clone().requires_grad_(True)
    ((F.conv2d(x, wt2, None, stride=1, padding=p) * dy).sum() + (F.conv2d(x2, wt2, None, stride=1, padding=p) * dy2).sum()).backward()
    ref2 = wt2.grad.permute(0, 2, 3, 1)
    for t in (0, 3, 6) if variant == 13 else (0, 1, 3):
        dw = torch.zeros(cout, k, k, cin, device="cuda")
        ops.conv2d_wgrad_grouped([(xx, dd), (to_nhwc(x2, torch.bfloat16), to_nhwc(dy2, torch.bfloat16))], dw, k, k, 1, p, cout, algo=1 + variant + 16 * t)
        assert (dw.cpu() - ref2).abs().max().item() <= 2e-2 * ref2.abs().max().item(), t


def test_conv_wgrad_ordered_mode_is_bit_reproducible():
    """Ordered mode (osd_conv_desc.ordered_ws, handed over with every call by ops.wgrad_set_workspace): with a scratch buffer registered for the stream the partial tiles are STORED and
    summed in a fixed order by a second launch.  For every launch form (single conv with ragged channel tiles, FPN levels
    sharing one dW, a mixed 1x1 / 3x3 / stride-2 launch with FrozenBN scales and bias gradients) and several variants /
    split targets: dW and db are BIT-IDENTICAL across repeats, equal the atomic path to rounding, and the atomic path
    itself is what differs from run to run; a workspace that is too small fails loudly."""
    from oneshotdet_amd import ops, _lib
    g = torch.Generator().manual_seed(3)
    mk = lambda *s: to_nhwc(torch.randn(*s, generator=g) * 0.5, torch.bfloat16)       # noqa: E731
    x1, dy1 = mk(3, 320, 40, 36), mk(3, 192, 40, 36)                                   # ragged: 320 / 192 channels
    lv = [(mk(2, 256, h, w), mk(2, 256, h, w)) for (h, w) in ((48, 64), (24, 32), (12, 16), (6, 8))]
    xm = [(mk(2, 256, 24, 32), mk(2, 128, 24, 32), 1, 1, 0), (mk(2, 128, 24, 32), mk(2, 128, 24, 32), 3, 1, 1),
          (mk(2, 256, 24, 32), mk(2, 512, 12, 16), 1, 2, 0)]
    scales = [(torch.rand(c, generator=g) + 0.5).cuda() for c in (128, 128, 512)]

    def run(algo):
        outs = []
        dw = torch.zeros(192, 3, 3, 320, device="cuda"); db = torch.zeros(192, device="cuda")
        ops.conv2d_wgrad(x1, dy1, dw, 3, 3, 1, 1, 192, db=db, algo=algo)
        outs += [dw, db]
        dw = torch.zeros(256, 3, 3, 256, device="cuda"); db = torch.zeros(256, device="cuda")
        ops.conv2d_wgrad_grouped(lv, dw, 3, 3, 1, 1, 256, db=db, algo=algo)
        outs += [dw, db]
        items = []
        for (x, dy, k, st, pd), sc in zip(xm, scales):
            cout = dy.shape[-1]
            items.append((x, dy, torch.zeros(cout, k, k, x.shape[-1], device="cuda"), sc, torch.zeros(cout, device="cuda"), k, k, st, pd, cout))
        ops.conv2d_wgrad_mixed(items, algo=algo)
        outs += [it[2] for it in items] + [it[4] for it in items]
        torch.cuda.synchronize()
        return outs
    algos = (1 + 0 + 16 * 4, 1 + 11 + 16 * 7, 1 + 5 + 16 * 0, 1 + 1 + 16 * 6)
    atomic = {a: run(a) for a in algos}
    try:
        ops.wgrad_set_workspace(nbytes=1 << 30)
        for a in algos:
            first = run(a)
            for _ in range(3):
                again = run(a)
                assert all(torch.equal(u, v) for u, v in zip(first, again)), a
            for u, v in zip(first, atomic[a]):
                assert (u - v).abs().max().item() <= 1e-4 * v.abs().max().item() + 1e-6, a
        ops.wgrad_set_workspace(nbytes=1 << 16)
        with pytest.raises(_lib.OsdError):
            run(algos[1])
    finally:
        ops.wgrad_set_workspace(nbytes=0)
    differs = any(not torch.equal(u, v) for a in algos for u, v in zip(run(a), atomic[a]))
    assert differs          # the default path adds with atomics: its rounding depends on the arrival order


@pytest.mark.parametrize("widths", [(32,), (64, 32, 16, 8), (128, 4), (96, 1)])
def test_conv_wgrad_filter_row_kernel_matches_autograd(widths):
    """conv_wgrad_xr_kernel (opt-in tuner candidate, OSD_WGRAD_XR=1; one workgroup per filter row: three taps on one dY fragment set, X tile read at row offsets
    0 / 1 / 2): every stage shape x several split targets against autograd — map widths that are multiples of the stage,
    equal to it, and NARROWER than it (zero-padded stages: P6 / P7), several levels adding into one dW, image borders
    (pad 1), the FrozenBN row scale and the fused bias gradient."""
    from oneshotdet_amd import ops
    cin, cout = 128, 256
    wt = (rnd(cout, cin, 3, 3, seed=2) / np.sqrt(cin * 9)).requires_grad_(True)
    b = torch.zeros(cout, requires_grad=True)
    scale = rnd(cout, seed=9).abs() + 0.5
    pairs, loss = [], 0
    for i, w in enumerate(widths):
        n, h = (2, 5) if w >= 16 else (3, 7)
        x = rnd(n, cin, h, w, seed=10 + i).bfloat16().float()
        dy = rnd(n, cout, h, w, seed=20 + i).bfloat16().float()
        loss = loss + (F.conv2d(x, wt * scale.view(-1, 1, 1, 1), b, padding=1) * dy).sum()
        pairs.append((to_nhwc(x, torch.bfloat16), to_nhwc(dy, torch.bfloat16)))
    loss.backward()
    ref, refb = wt.grad.permute(0, 2, 3, 1), b.grad
    cands = ops.wgrad_xr_candidates(ops.OSD_BF16, cout, cin, 3, 3, 1, 1, list(widths))
    assert cands and all(a > 128 for a in cands)
    if widths == (96, 1):
        assert {(a - 129) & 15 for a in cands} == {0, 2}          # 96 is no multiple of the 64-pixel stage
    for algo in cands:
        dw, db = torch.zeros(cout, 3, 3, cin, device="cuda"), torch.zeros(cout, device="cuda")
        ops.conv2d_wgrad_grouped(pairs, dw, 3, 3, 1, 1, cout, scale=scale.cuda(), db=db, algo=algo)
        assert (dw.cpu() - ref).abs().max().item() <= 2e-2 * ref.abs().max().item(), algo
        assert (db.cpu() - refb).abs().max().item() <= 2e-2 * refb.abs().max().item(), algo
    assert not ops.wgrad_xr_candidates(ops.OSD_BF16, cout, cin, 3, 3, 2, 1, [32])       # stride 2: not this kernel
    assert not ops.wgrad_xr_candidates(ops.OSD_BF16, cout, 64, 3, 3, 1, 1, [32])        # channels in 128s
    with pytest.raises(Exception):
        ops.conv2d_wgrad(pairs[0][0][..., :64].contiguous(), pairs[0][1], torch.zeros(cout, 3, 3, 64, device="cuda"), 3, 3, 1, 1, cout,
                         algo=129)


@pytest.mark.parametrize("dt", ["f32", "bf16"])
def test_conv_wgrad_batched_equals_one_launch_per_conv(dt):
    """osd_conv2d_wgrad_batched: k convs of identical geometry with their own weights / FrozenBN scales / bias gradients
    in one launch vs k launches of osd_conv2d_wgrad (accumulation order differs only through the atomics)."""
    from oneshotdet_amd import ops
    n, cin, h, w, cout, k, stride, pad = 2, 256, 14, 18, 128, 3, 1, 1
    items, refs = [], []
    for i in range(5):
        x = to_nhwc(rnd(n, cin, h, w, seed=10 + i), DT[dt])
        dy = to_nhwc(rnd(n, cout, h, w, seed=20 + i), DT[dt])
        scale = (rnd(cout, seed=30 + i).abs() + 0.5).cuda() if i % 2 == 0 else None
        dw, db = torch.zeros(cout, k, k, cin, device="cuda"), (torch.zeros(cout, device="cuda") if i != 1 else None)
        rw, rb = torch.zeros_like(dw), (torch.zeros(cout, device="cuda") if i != 1 else None)
        ops.conv2d_wgrad(x, dy, rw, k, k, stride, pad, cout, scale=scale, db=rb)
        items.append((x, dy, dw, scale, db))
        refs.append((rw, rb))
    for algo in (None, 1 + 0 + 16 * 3, 1 + 1 + 16 * 0):
        for it in items:
            it[2].zero_()
            if it[4] is not None:
                it[4].zero_()
        ops.conv2d_wgrad_batched(items, k, k, stride, pad, cout, algo=algo)
        for (x, dy, dw, scale, db), (rw, rb) in zip(items, refs):
            torch.testing.assert_close(dw, rw, rtol=1e-3, atol=1e-3 * float(rw.abs().max()))
            if db is not None:
                torch.testing.assert_close(db, rb, rtol=1e-3, atol=1e-3 * float(rb.abs().max()))


def test_groupnorm_relu_backward_with_zero_negative_and_tiny_scales():
    """gamma == 0, gamma < 0 and |gamma| ~ 1e-6 are legal GroupNorm parameters (nothing in the reference's optimiser keeps
    them positive): the backward pass takes xhat from the saved normalisation, never from (z - beta) / gamma, so the
    gradients stay finite and equal to autograd's — including d gamma of the channels whose gamma is exactly zero."""
    from oneshotdet_amd import ops
    g = rnd(256, seed=3) * 0.7
    g[::7] = 0.0
    g[1::11] = 1e-6
    g[2::13] = -0.4
    g.requires_grad_(True)
    b = rnd(256, seed=4).requires_grad_(True)
    sizes = [(25, 32), (7, 8)]
    xs, dys = [], []
    for i, (h, w) in enumerate(sizes):
        x = (rnd(2, 256, h, w, seed=10 + i, scale=2) + 0.3).requires_grad_(True)
        dy = rnd(2, 256, h, w, seed=20 + i)
        (F.relu(F.group_norm(x, 32, g, b, eps=1e-5)) * dy).sum().backward()
        xs.append(x); dys.append(dy)
    xx = [to_nhwc(x.detach(), torch.float32) for x in xs]
    _, ab = ops.groupnorm_relu_levels(xx, g.detach().cuda(), b.detach().cuda(), 32, 1e-5)
    dg, db = torch.zeros(256, device="cuda"), torch.zeros(256, device="cuda")
    dus = ops.groupnorm_relu_bwd_levels(xx, [to_nhwc(d, torch.float32) for d in dys], ab, g.detach().cuda(), b.detach().cuda(),
                                        dg, db, 32)
    for x, du in zip(xs, dus):
        got = du.cpu().permute(0, 3, 1, 2)
        assert torch.isfinite(got).all()
        assert (got - x.grad).abs().max() <= 2e-4 * x.grad.abs().max()
    assert (dg.cpu() - g.grad).abs().max() <= 2e-4 * g.grad.abs().max()
    assert g.grad[::7].abs().max() > 0 and (dg.cpu()[::7] - g.grad[::7]).abs().max() <= 2e-4 * g.grad.abs().max()
    assert (db.cpu() - b.grad).abs().max() <= 2e-4 * b.grad.abs().max()


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("n", [1, 2])
def test_groupnorm_relu_levels_forward_backward_at_fpn_sizes(n, dt):
    """The level-grouped GroupNorm+ReLU (2 launches forward, 2 backward over P3..P7) at the BASELINE map sizes
    (100x128 ... 7x8) against torch autograd: outputs, input gradients and the shared d gamma / d beta."""
    from oneshotdet_amd import ops
    sizes = [(100, 128), (50, 64), (25, 32), (13, 16), (7, 8)]
    g = (rnd(256, seed=3).abs() + 0.5).requires_grad_(True)
    b = rnd(256, seed=4).requires_grad_(True)
    xs, dys, refs = [], [], []
    for i, (h, w) in enumerate(sizes):
        x = rnd(n, 256, h, w, seed=10 + i, scale=2) + 0.3
        dy = rnd(n, 256, h, w, seed=20 + i) * (10.0 if i == 1 else 1.0)
        if dt == "bf16":
            x, dy = x.bfloat16().float(), dy.bfloat16().float()
        x.requires_grad_(True)
        y = F.relu(F.group_norm(x, 32, g, b, eps=1e-5))
        (y * dy).sum().backward()
        xs.append(x); dys.append(dy); refs.append(y.detach())
    xx = [to_nhwc(x.detach(), DT[dt]) for x in xs]
    ys, ab = ops.groupnorm_relu_levels(xx, g.detach().cuda(), b.detach().cuda(), 32, 1e-5)
    dg, db = torch.zeros(256, device="cuda"), torch.zeros(256, device="cuda")
    dus = ops.groupnorm_relu_bwd_levels(xx, [to_nhwc(d, DT[dt]) for d in dys], ab, g.detach().cuda(), b.detach().cuda(), dg, db, 32)
    tol = 2e-4 if dt == "f32" else 5e-2
    for x, y, du, ref in zip(xs, ys, dus, refs):
        assert (y.float().cpu().permute(0, 3, 1, 2) - ref).abs().max() <= tol * ref.abs().max(), x.shape
        gref = x.grad
        assert (du.float().cpu().permute(0, 3, 1, 2) - gref).abs().max() <= tol * gref.abs().max(), x.shape
    assert (dg.cpu() - g.grad).abs().max() <= tol * g.grad.abs().max()
    assert (db.cpu() - b.grad).abs().max() <= tol * b.grad.abs().max()


@pytest.mark.parametrize("dt", ["f32", "bf16"])
def test_groupnorm_backward_also_gives_the_producing_convs_bias_gradient(dt):
    """osd_groupnorm_relu_bwd_levels_convbias: the bias gradient of the conv in front of the GroupNorm (fcos.py:29-37,
    Conv2d(bias=True) -> GroupNorm -> ReLU) is sum_px du, which the kernels get from their own sums (gamma sum dz, the group means,
    sum xhat) without a pass over du.  Against autograd's x.grad summed over images and pixels, and against the du the same call
    wrote; du / d gamma / d beta are unchanged by the extra output.  gamma == 0 and gamma < 0 channels, five FPN sizes, two images."""
    from oneshotdet_amd import ops
    sizes = [(100, 128), (50, 64), (25, 32), (13, 16), (7, 8)]
    g = rnd(256, seed=3) * 0.7
    g[::7] = 0.0
    g[2::13] = -0.4
    g.requires_grad_(True)
    b = rnd(256, seed=4).requires_grad_(True)
    xs, dys = [], []
    for i, (h, w) in enumerate(sizes):
        x = rnd(2, 256, h, w, seed=10 + i, scale=2) + 0.3
        dy = rnd(2, 256, h, w, seed=20 + i) * (10.0 if i == 1 else 1.0)
        if dt == "bf16":
            x, dy = x.bfloat16().float(), dy.bfloat16().float()
        x.requires_grad_(True)
        (F.relu(F.group_norm(x, 32, g, b, eps=1e-5)) * dy).sum().backward()
        xs.append(x); dys.append(dy)
    ref_db = sum(x.grad.sum(dim=(0, 2, 3)) for x in xs)
    xx = [to_nhwc(x.detach(), DT[dt]) for x in xs]
    dd = [to_nhwc(d, DT[dt]) for d in dys]
    gc, bc = g.detach().cuda(), b.detach().cuda()
    _, ab = ops.groupnorm_relu_levels(xx, gc, bc, 32, 1e-5)
    dg0, db0 = torch.zeros(256, device="cuda"), torch.zeros(256, device="cuda")
    du0 = ops.groupnorm_relu_bwd_levels(xx, dd, ab, gc, bc, dg0, db0, 32)
    dg1, db1, cdb = torch.zeros(256, device="cuda"), torch.zeros(256, device="cuda"), torch.zeros(256, device="cuda")
    du1 = ops.groupnorm_relu_bwd_levels(xx, dd, ab, gc, bc, dg1, db1, 32, conv_db=cdb)
    assert all(torch.equal(p, q) for p, q in zip(du0, du1))
    torch.testing.assert_close(dg1, dg0, rtol=1e-5, atol=1e-5 * float(dg0.abs().max()))
    torch.testing.assert_close(db1, db0, rtol=1e-5, atol=1e-5 * float(db0.abs().max()))
    own = sum(d.float().sum(dim=(0, 1, 2)) for d in du1)
    scale = float(ref_db.abs().max())
    tol = 2e-4 if dt == "f32" else 5e-3          # bf16: autograd sees fp32 du, `own` their bf16 roundings; the kernel's sum is of the fp32 du
    assert float((cdb.cpu() - ref_db).abs().max()) <= tol * scale
    # `own` sums the STORED du: in bf16 each carries a rounding of up to 2^-9 relative, and the roundings of a channel are not
    # independent (a gamma == 0 channel stores almost the same value at every pixel): bound them by their worst case; the kernel's
    # value is the sum of the unrounded du, which the autograd comparison above pins
    worst = sum(d.float().abs().sum(dim=(0, 1, 2)) for d in du1) * 2.0 ** -9
    bound = 2e-4 * scale + (worst if dt == "bf16" else 0.0)
    assert bool(((cdb - own).abs() <= bound).all()), float(((cdb - own).abs() / bound).max())
    # accumulates: a second call adds the same again
    ops.groupnorm_relu_bwd_levels(xx, dd, ab, gc, bc, dg1, db1, 32, conv_db=cdb)
    assert float((cdb.cpu() - 2 * ref_db).abs().max()) <= 2 * tol * scale


@pytest.mark.parametrize("fused_levels", [2, 1])
def test_groupnorm_backward_statistics_gathered_by_the_data_gradient_conv(fused_levels):
    """osd_conv2d_fwd_multi_gn + osd_groupnorm_relu_bwd_levels_fused: the 3x3 data-gradient conv that writes dt (the gradient
    w.r.t. a GroupNorm + ReLU output) also gathers that GroupNorm's backward sums in its epilogue, and the GroupNorm backward
    then skips its statistics pass for those levels.  Two towers x two levels (4 x 64 and 2 x 64 maps of 2 images: whole
    256-pixel tiles, whole 128-pixel runs per image) with separate weights and GroupNorm parameters, a zero and a negative
    gamma, with both levels or only the first one gathered: dt is bit-identical to the plain launch, du equal up to the
    summation order of the sums (<= 1 bf16 ulp on a few elements), d gamma / d beta to 1e-4 relative."""
    from oneshotdet_amd import ops
    n, c, G = 2, 256, 32
    sizes = [(4, 64), (2, 64)]
    xs, us, pcs, gammas, betas = [], [], [], [], []
    for tw in range(2):
        wt = rnd(c, c, 3, 3, seed=40 + tw) / np.sqrt(c * 9)
        pcs.append(ops.pack_conv(wt.cuda(), bias=torch.zeros(c).cuda(), dtype=torch.bfloat16))
        g = rnd(c, seed=50 + tw)
        g[3], g[100] = 0.0, -0.7
        gammas.append(g.cuda())
        betas.append((rnd(c, seed=60 + tw) * 0.3).cuda())
    seg_x, seg_u, seg_pc, seg_tw = [], [], [], []
    for l, (h, w) in enumerate(sizes):          # segment order of the engine: level-major, towers inside
        for tw in range(2):
            seg_x.append(to_nhwc(rnd(n, c, h, w, seed=70 + 2 * l + tw), torch.bfloat16))
            seg_u.append(to_nhwc(rnd(n, c, h, w, seed=80 + 2 * l + tw, scale=2) + 0.3, torch.bfloat16))
            seg_pc.append(pcs[tw]); seg_tw.append(tw)
    abs_ = []
    for tw in range(2):
        _, ab = ops.groupnorm_relu_levels([seg_u[2 * l + tw] for l in range(2)], gammas[tw], betas[tw], G, 1e-5)
        abs_.append(ab)
    # plain: conv, then the two-pass GroupNorm backward
    k = 2 * fused_levels      # the gathering launch covers the leading segments, the rest goes out as a plain launch
    dts = ops.conv2d_multi(seg_x[:k], seg_pc[:k], pad=1, algo=ops.ALGO_SP, _whole=True)
    if k < len(seg_x):
        dts += ops.conv2d_multi(seg_x[k:], seg_pc[k:], pad=1)
    ref = []
    for tw in range(2):
        dg, db = torch.zeros(c, device="cuda"), torch.zeros(c, device="cuda")
        dus = ops.groupnorm_relu_bwd_levels([seg_u[2 * l + tw] for l in range(2)], [dts[2 * l + tw] for l in range(2)], abs_[tw],
                                            gammas[tw], betas[tw], dg, db, G)
        ref.append((dus, dg, db))
    # gathered: zeroed workspaces, the conv with the statistics arguments, the GroupNorm backward with the levels' bits set
    wss = [torch.zeros(ops.gn_bwd_ws_numel(2, n, c, G), device="cuda") for _ in range(2)]
    gnb = {"us": [], "abs": [], "gammas": [], "wss": [], "pws": [], "n": n, "groups": G}
    for l in range(2):
        for tw in range(2):
            on = l < fused_levels
            ws_l, pw_l = ops.gn_bwd_ws_parts(wss[tw], 2, n, c, G)[l]
            gnb["us"].append(seg_u[2 * l + tw] if on else None)
            gnb["abs"].append(abs_[tw][l] if on else None)
            gnb["gammas"].append(gammas[tw] if on else None)
            gnb["wss"].append(ws_l if on else None)
            gnb["pws"].append(pw_l if on else None)
    dts2 = ops.conv2d_multi(seg_x, seg_pc, pad=1, gnb=gnb)
    for a, b in zip(dts, dts2):
        assert torch.equal(a, b)
    for tw in range(2):
        dg, db = torch.zeros(c, device="cuda"), torch.zeros(c, device="cuda")
        dus = ops.groupnorm_relu_bwd_levels([seg_u[2 * l + tw] for l in range(2)], [dts2[2 * l + tw] for l in range(2)], abs_[tw],
                                            gammas[tw], betas[tw], dg, db, G, ws=wss[tw], fused_mask=(1 << fused_levels) - 1)
        rdus, rdg, rdb = ref[tw]
        for a, b in zip(dus, rdus):
            a, b = a.float(), b.float()
            assert (a - b).abs().max() <= 2 ** -7 * b.abs().max()
            assert (a != b).float().mean() <= 1e-3
        torch.testing.assert_close(dg, rdg, rtol=1e-4, atol=1e-4 * float(rdg.abs().max()))
        torch.testing.assert_close(db, rdb, rtol=1e-4, atol=1e-4 * float(rdb.abs().max()))


def test_groupnorm_forward_statistics_gathered_by_the_tower_conv():
    """osd_conv2d_fwd_multi_gn (forward statistics) + osd_groupnorm_relu_fwd_levels_fused: the tower conv's epilogue adds the sum
    and the sum of squares of the outputs it stores to the GroupNorm's slab sums; the GroupNorm skips its statistics pass for
    those levels.  Two towers x three levels (the third too small to qualify: it keeps the statistics pass): conv outputs
    bit-identical, GroupNorm outputs equal up to the summation order of the sums (<= 1 bf16 ulp on a few elements), the saved
    scale / shift (ab) to 1e-5 relative."""
    from oneshotdet_amd import ops
    n, c, G = 2, 256, 32
    sizes = [(4, 64), (2, 64), (3, 5)]
    pcs = [ops.pack_conv((rnd(c, c, 3, 3, seed=40 + tw) / np.sqrt(c * 9)).cuda(), bias=(rnd(c, seed=45 + tw) * 0.2).cuda(), dtype=torch.bfloat16)
           for tw in range(2)]
    gam = [rnd(c, seed=50 + tw).cuda() for tw in range(2)]
    bet = [(rnd(c, seed=60 + tw) * 0.3).cuda() for tw in range(2)]
    seg_x = [to_nhwc(rnd(n, c, h, w, seed=70 + 2 * l + tw), torch.bfloat16) for l, (h, w) in enumerate(sizes) for tw in range(2)]
    seg_pc = [pcs[tw] for _ in sizes for tw in range(2)]
    nl, nf = len(sizes), 2
    us = ops.conv2d_multi(seg_x[:2 * nf], seg_pc[:2 * nf], pad=1, algo=ops.ALGO_SP, _whole=True) + ops.conv2d_multi(seg_x[2 * nf:], seg_pc[2 * nf:], pad=1)
    ref = [ops.groupnorm_relu_levels(us[tw::2], gam[tw], bet[tw], G, 1e-5) for tw in range(2)]
    wss = [torch.zeros(nl * n * ops.GN_SPLITS * G * 2, device="cuda") for _ in range(2)]
    parts = [ops.gn_fwd_ws_parts(wss[tw], nl, n, G) for tw in range(2)]
    gnb = {"wss": [parts[tw][l] if l < nf else None for l in range(nl) for tw in range(2)], "n": n, "groups": G}
    us2 = ops.conv2d_multi(seg_x, seg_pc, pad=1, gnb=gnb)
    for a, b in zip(us, us2):
        assert torch.equal(a, b)
    for tw in range(2):
        ys, ab = ops.groupnorm_relu_levels(us2[tw::2], gam[tw], bet[tw], G, 1e-5, ws=wss[tw], fused_mask=(1 << nf) - 1)
        rys, rab = ref[tw]
        torch.testing.assert_close(ab, rab, rtol=1e-5, atol=1e-5)
        for a, b in zip(ys, rys):
            a, b = a.float(), b.float()
            assert (a - b).abs().max() <= 2 ** -7 * b.abs().max()
            assert (a != b).float().mean() <= 1e-3


def test_training_step_with_gathered_groupnorm_statistics_equals_the_two_pass_step():
    """TrainEngine.fuse_gn_bwd (OSD_GN_FUSION=1; off by default): at the BASELINE geometry (800x1024: P3 and P4
    qualify) the tower data-gradient convs of layers 3..1 gather the GroupNorm-backward sums of layers 2..0 for P3 + P4.  The
    forward pass is untouched and the backward pass is linear in the loss gradient, so the parameter gradients of the two
    engines differ only by the summation order of those sums: the head's gradients to 1e-4 relative L2 (measured 3e-6), the backbones'
    — behind ~50 layers that re-round the slightly different data gradients to bf16 — to 1e-2 (measured 8e-4 .. 5e-3)."""
    from oneshotdet_amd import _lib, ops
    e0, img, q, gtb, cnt = _engine_and_inputs("bf16", "config1")
    img, q, gtb, cnt = (t.expand(2, *t.shape[1:]).contiguous() for t in (img, q, gtb, cnt))
    assert not e0.fuse_gn_bwd
    e0.fuse_gn_fwd = False      # the forward statistics too: both engines then share ONE forward pass bit for bit
    l0 = e0.forward_backward(img, q, gtb, cnt).clone()
    g0 = e0.flat_g.clone()
    e0.fuse_gn_bwd = True
    calls = []
    real = _lib.call

    def spy(name, *a):
        calls.append(name)
        return real(name, *a)
    _lib.call = ops._lib.call = spy
    try:
        l1 = e0.forward_backward(img, q, gtb, cnt).clone()
    finally:
        _lib.call = ops._lib.call = real
    assert calls.count("osd_conv2d_fwd_multi_gn") == 2 * (spec.NUM_CONVS - 1)      # one chain per tower, layers 3..1
    torch.testing.assert_close(l0, l1, rtol=1e-5, atol=0)      # the loss sums are atomic adds: equal up to their order
    errs = {name: float((e0.flat_g[lo:hi] - g0[lo:hi]).norm() / g0[lo:hi].norm()) for name, (lo, hi) in e0.exchange.ranges.items()}
    print("\ngathered vs two-pass GroupNorm statistics, relative L2 per gradient bucket:", {k: "%.1e" % v for k, v in errs.items()})
    assert errs["head"] <= 1e-4 and max(errs.values()) <= 1e-2, errs


@pytest.mark.parametrize("dt", ["f32", "bf16"])
def test_full_size_batch8_training_step_properties(dt):
    """BASELINE.json configs[2] size (8 x 800x1024 targets, 8 x 127x127 queries).  A batch of 8 identical (image, query,
    boxes) triples must give the single-image losses (the FCOS losses are normalised by the total number of positives)
    and the single-image parameter gradients — and the single-image fp32 run is pinned to the reference by
    test_forward_backward_matches_oracle_autograd[config1].  Exercises every training kernel at the benchmark's grid
    sizes: batched / grouped weight gradients with many pixel splits, level-grouped GroupNorm, the loss kernels."""
    from oneshotdet_amd import ops
    e1, img, q, gtb, cnt = _engine_and_inputs(dt, "config1")
    # both batch sizes with the library's default kernels: whatever an earlier test's ops.tuning() cached for ONE of the two batch
    # sizes (e.g. the row-reuse family, which sums K in another order) would make the comparison measure bf16's sensitivity to
    # the summation order (3 % at the deepest layers) instead of batching invariance
    saved = {name: dict(getattr(ops, name)) for name in ("ALGO_CACHE", "SPLIT_CACHE", "WGRAD_ALGO_CACHE")}
    for name in saved:
        getattr(ops, name).clear()
    try:
        l1 = e1.forward_backward(img, q, gtb, cnt).clone()
        g1 = e1.flat_g.clone()
        l8 = e1.forward_backward(img.expand(8, -1, -1, -1).contiguous(), q.expand(8, -1, -1, -1).contiguous(),
                                 gtb.expand(8, -1, -1).contiguous(), cnt.expand(8).contiguous()).clone()
    finally:
        for name, d in saved.items():
            getattr(ops, name).update(d)
    g8 = e1.flat_g
    assert int(l8[3]) == 8 * int(l1[3])
    torch.testing.assert_close(l8[:3], l1[:3], rtol=1e-5 if dt == "f32" else 2e-3, atol=0)
    for name, (lo, hi) in e1.exchange.ranges.items():
        a, b = g8[lo:hi], g1[lo:hi]
        err = float((a - b).norm() / b.norm())
        assert err <= (2e-4 if dt == "f32" else 3e-2), (name, err)


def _batch_of(name, order, dt):
    """A batch assembled from the images of fixture case `name` in the given order (indices into the case's batch)."""
    from oneshotdet_amd import train
    B, H, W, S, qh, qw = gu.CASES[name]
    img, q = gu.case_inputs(name)
    gts = synth.make_gt_boxes(B, H, W, seed=3, max_boxes=3)
    G = max(len(g) for g in gts)
    gtb = torch.zeros(len(order), G, 4)
    for j, i in enumerate(order):
        gtb[j, :len(gts[i])] = torch.from_numpy(gts[i])
    cnt = torch.tensor([len(gts[i]) for i in order], dtype=torch.int32)
    eng = train.TrainEngine(synth.make_state_dict(spec.hot_path_shapes()), dtype=DT[dt])
    idx = torch.tensor(order)
    return eng, torch.from_numpy(img)[idx].cuda(), torch.from_numpy(q)[idx].cuda(), gtb.cuda(), cnt.cuda()


@pytest.mark.parametrize("dt", ["f32", "bf16"])
def test_full_size_batch8_of_two_distinct_images_matches_the_reference_fixture(dt):
    """bs=8 at 800x1024 built from the two DISTINCT (image, query, boxes) triples of train_config1x2.npz (recorded through
    the reference at batch 2) in an irregular order [A,B,B,A,B,A,A,B].  The FCOS losses are normalised over the whole
    batch (fcos/loss.py:251-271), so a batch with every triple four times has the fixture's losses and parameter gradients:
    image-index mistakes in any forward / loss / data-gradient / weight-gradient kernel at the benchmark's grid sizes show
    as a difference between A-slots and B-slots.  The training proposals of every slot equal those of the other slots
    with the same image, and differ between A and B."""
    f = gu.load("train_config1x2.npz")
    order = [0, 1, 1, 0, 1, 0, 0, 1]
    eng, img, q, gtb, cnt = _batch_of("config1x2", order, dt)
    losses = eng.forward_backward(img, q, gtb, cnt).cpu().numpy()
    assert int(losses[3]) == 4 * int(f["num_pos"])
    np.testing.assert_allclose(losses[:3], f["losses_cuda_formula"], rtol=1e-4 if dt == "f32" else 3e-2)
    _check_grads_against_fixture(eng.named_grads(), f, dt)
    torch.cuda.synchronize()
    pb, ps, pc = eng.proposals
    for j in range(2, 8):
        k = order.index(order[j])
        assert int(pc[j]) == int(pc[k]) and torch.equal(pb[j], pb[k]) and torch.equal(ps[j], ps[k]), (j, k)
    assert not torch.equal(pb[0], pb[1])


@pytest.mark.parametrize("dt", ["f32", "bf16"])
def test_full_size_batch8_of_eight_distinct_images_has_additive_loss_sums(dt):
    """bench.py's batch: 8 DISTINCT images, queries and box sets.  The un-normalised loss sums {num_pos, sum_w, sum_focal,
    sum_w*(1-GIoU), sum_bce} of the batch are the sums of the eight single-image runs (per-image target assignment and
    loss terms do not see the other images), and the batch losses follow from them by fcos/loss.py:251-271."""
    from oneshotdet_amd import train
    B = 8
    images = torch.from_numpy(synth.make_images("bench.target", B, 800, 1024, seed=1000)).cuda()
    queries = torch.from_numpy(synth.make_images("bench.query", B, 127, 127, seed=1000)).cuda()
    gts = synth.make_gt_boxes(B, 800, 1024, seed=1000, max_boxes=6)
    gtb = torch.zeros(B, 6, 4)
    for i, g in enumerate(gts):
        gtb[i, :len(g)] = torch.from_numpy(g)
    gtb = gtb.cuda()
    cnt = torch.tensor([len(g) for g in gts], dtype=torch.int32).cuda()
    eng = train.TrainEngine(synth.make_state_dict(spec.hot_path_shapes()), dtype=DT[dt])
    l8 = eng.forward_backward(images, queries, gtb, cnt, with_proposals=False).cpu().double()
    s8 = eng.last_loss_sums[:5].cpu().double()
    tot = torch.zeros(5, dtype=torch.float64)
    for i in range(B):
        eng.forward_backward(images[i:i + 1].contiguous(), queries[i:i + 1].contiguous(), gtb[i:i + 1].contiguous(),
                             cnt[i:i + 1].contiguous(), with_proposals=False)
        tot += eng.last_loss_sums[:5].cpu().double()
    assert int(s8[0]) == int(tot[0]) and int(tot[0]) > 8
    # the forward is batch-invariant bit for bit (test_gpu_parity); the sums differ only by the order of the atomics
    torch.testing.assert_close(s8, tot, rtol=1e-5, atol=1e-6)
    expect = torch.stack([tot[2] / (tot[0] + B), tot[3] / tot[1], tot[4] / tot[0]])
    torch.testing.assert_close(l8[:3], expect, rtol=1e-5, atol=1e-7)


def test_conv_wgrad_multi_mixes_shared_and_own_weights():
    """osd_conv2d_wgrad_multi: pairs of different spatial size, some sharing a dW (FPN levels of one conv), others with
    their own (another conv of the tower), against one osd_conv2d_wgrad launch per pair."""
    from oneshotdet_amd import ops
    cin = cout = 256
    sizes = [(2, 25, 32), (2, 13, 16), (2, 7, 8)]
    dws = [torch.zeros(cout, 3, 3, cin, device="cuda") for _ in range(2)]
    dbs = [torch.zeros(cout, device="cuda") for _ in range(2)]
    refs = [torch.zeros_like(dws[0]) for _ in range(2)]
    refb = [torch.zeros_like(dbs[0]) for _ in range(2)]
    items = []
    for conv in range(2):
        for i, (n, h, w) in enumerate(sizes):
            x = to_nhwc(rnd(n, cin, h, w, seed=100 * conv + i), torch.bfloat16)
            dy = to_nhwc(rnd(n, cout, h, w, seed=100 * conv + 50 + i), torch.bfloat16)
            items.append((x, dy, dws[conv], None, dbs[conv]))
            ops.conv2d_wgrad(x, dy, refs[conv], 3, 3, 1, 1, cout, db=refb[conv])
    for algo in (None, 1 + 0 + 16 * 2, 1 + 4 + 16 * 0):
        for t in dws + dbs:
            t.zero_()
        ops.conv2d_wgrad_multi(items, 3, 3, 1, 1, cout, algo=algo)
        for conv in range(2):
            torch.testing.assert_close(dws[conv], refs[conv], rtol=1e-3, atol=1e-3 * float(refs[conv].abs().max()))
            torch.testing.assert_close(dbs[conv], refb[conv], rtol=1e-3, atol=1e-3 * float(refb[conv].abs().max()))


@pytest.mark.parametrize("dt", ["f32", "bf16"])
def test_conv_wgrad_mixed_geometries_in_one_launch(dt):
    """osd_conv2d_wgrad_mixed: 1x1 and 3x3, stride 1 and 2, different channel counts and map sizes, FrozenBN scales and
    bias gradients — one launch vs one osd_conv2d_wgrad launch per conv."""
    from oneshotdet_amd import ops
    cases = [(2, 256, 14, 18, 128, 1, 1, 0), (2, 128, 14, 18, 128, 3, 1, 1), (2, 128, 14, 18, 512, 1, 1, 0),
             (1, 512, 16, 12, 256, 1, 2, 0), (2, 256, 7, 9, 256, 3, 2, 1), (3, 64, 5, 4, 320, 3, 1, 1)]
    items, refs = [], []
    for i, (n, cin, h, w, cout, k, stride, pad) in enumerate(cases):
        x = to_nhwc(rnd(n, cin, h, w, seed=10 + i), DT[dt])
        ho, wo = ops.conv_out(h, k, stride, pad), ops.conv_out(w, k, stride, pad)
        dy = to_nhwc(rnd(n, cout, ho, wo, seed=20 + i), DT[dt])
        scale = (rnd(cout, seed=30 + i).abs() + 0.5).cuda() if i % 2 == 0 else None
        dw = torch.zeros(cout, k, k, cin, device="cuda")
        db = torch.zeros(cout, device="cuda") if i % 3 == 0 else None
        rw, rb = torch.zeros_like(dw), (torch.zeros(cout, device="cuda") if db is not None else None)
        ops.conv2d_wgrad(x, dy, rw, k, k, stride, pad, cout, scale=scale, db=rb)
        items.append((x, dy, dw, scale, db, k, k, stride, pad, cout))
        refs.append((rw, rb))
    algos = (None, 1 + 0 + 16 * 3, 1 + 0 + 16 * 4) + ((1 + 1 + 16 * 0, 1 + 4 + 16 * 1, 1 + 8, 1 + 10) if dt == "bf16" else ())
    for algo in algos:
        for it in items:
            it[2].zero_()
            if it[4] is not None:
                it[4].zero_()
        ops.conv2d_wgrad_mixed(items, algo=algo)
        for it, (rw, rb) in zip(items, refs):
            torch.testing.assert_close(it[2], rw, rtol=1e-3, atol=1e-3 * float(rw.abs().max()))
            if rb is not None:
                torch.testing.assert_close(it[4], rb, rtol=1e-3, atol=1e-3 * float(rb.abs().max()))


def test_conv_wgrad_owner_mode_adds_into_a_nonzero_gradient_and_respects_shared_weights():
    """Round 6: a segment whose pixels are ONE split owns its dW tiles and adds them with plain loads + stores instead of memory-side
    atomics (wgrad_params.h: wg_owner_add) — the `+=` contract must hold on a gradient buffer that is NOT zero, and two segments
    that name the SAME dW (FPN levels sharing a conv) must keep the atomics (each would overwrite the other's sum).  Every tile family: the 128 x 128 ring kernel, the 256 x 256 8-wave tiles, the
    software-pipelined kernel with the launcher's splits; full and ragged channel tiles."""
    from oneshotdet_amd import ops
    for (n, cin, h, w, cout, k) in [(2, 256, 20, 20, 512, 3), (1, 320, 20, 26, 384, 1), (2, 512, 10, 16, 256, 1)]:
        p = k // 2
        x = rnd(n, cin, h, w, seed=41).bfloat16().float()
        dy = rnd(n, cout, h, w, seed=42).bfloat16().float()
        wt = (rnd(cout, cin, k, k, seed=43) / np.sqrt(cin * k * k)).requires_grad_(True)
        (F.conv2d(x, wt, None, stride=1, padding=p) * dy).sum().backward()
        ref = wt.grad.permute(0, 2, 3, 1)
        xx, dd = to_nhwc(x, torch.bfloat16), to_nhwc(dy, torch.bfloat16)
        init = rnd(cout, k, k, cin, seed=44).cuda() * float(ref.abs().max())
        for algo in (1 + 0 + 16 * 3, 1 + 4 + 16 * 3, 1 + 5 + 16 * 3, 1 + 13 + 16 * 3, 1 + 13 + 16 * 2, 1 + 10 + 16 * 3, 1 + 11 + 16 * 3):
            dw = init.clone()
            ops.conv2d_wgrad(xx, dd, dw, k, k, 1, p, cout, algo=algo)             # target 64 / 128 workgroups: one split per tile
            assert ((dw - init).cpu() - ref).abs().max().item() <= 2e-2 * ref.abs().max().item(), (n, cin, cout, k, algo)
            # the same tensors as two "levels" of one conv: shared dW -> atomics, twice the gradient on top of the initial values
            dw = init.clone()
            ops.conv2d_wgrad_grouped([(xx, dd), (xx, dd)], dw, k, k, 1, p, cout, algo=algo)
            assert ((dw - init).cpu() - 2 * ref).abs().max().item() <= 2e-2 * 2 * ref.abs().max().item(), (n, cin, cout, k, algo)


def test_two_ranks_on_one_gpu_average_gradients():
    """World size 2 with BOTH ranks on this GPU (gloo moves the device tensors): the real TrainEngine with the
    overlapped, bucketed gradient exchange must produce the average of the two ranks' gradients, and train_step must
    apply the same update on both ranks.  (The RCCL flavour of the same path runs with one rank in
    test_gradient_buckets_cover_the_flat_buffer_and_overlapped_exchange_runs; 8 GPUs are the driver's.)"""
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "tests", "dist_gpu_worker.py")]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=root,
                         env=dict(os.environ, OMP_NUM_THREADS="4"))
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "RANK 0 GPU_EXCHANGE=True" in out.stdout and "RANK 1 GPU_EXCHANGE=True" in out.stdout, out.stdout[-1500:]


def test_bench_two_ranks_sharing_the_gpu():
    """`bench.py --gpus 2` exactly as the driver launches it (torch.distributed.run, one process per rank), with both ranks
    placed on this GPU (OSD_BENCH_SHARE_GPU=1: gloo instead of RCCL): one JSON line from rank 0, whole-job value = 16
    images per step / MAX-over-ranks time, the exchange in the parallelism string."""
    import json
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--no-conv-timing"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=root,
                         env=dict(os.environ, OSD_BENCH_SHARE_GPU="1", OMP_NUM_THREADS="4"))
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["config"]["global_batch"] == 16 and r["scaling"] == "weak"
    assert abs(r["value"] - 16 * 1e3 / r["ms_per_step"]) / r["value"] < 0.01
    assert "buckets" in r["config"]["parallelism"] and "cpu_baseline" not in r


def test_bench_self_launches_two_ranks_sharing_the_gpu():
    """`python3 bench.py --gpus 2` with NO launcher (no WORLD_SIZE): bench.py starts its own two ranks before touching
    the GPU; both share this GPU (OSD_BENCH_SHARE_GPU=1, gloo).  The line reports n_gpus 2 and the process group's own
    world size; a single-rank run says there is no collective."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(OSD_BENCH_SHARE_GPU="1", OMP_NUM_THREADS="4")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                          "--no-conv-timing"], capture_output=True, text=True, timeout=900, cwd=root, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["config"]["world_size"] == 2 and r["config"]["global_batch"] == 16
    assert "buckets" in r["config"]["parallelism"] and "gloo" in r["config"]["parallelism"]
    # --gpus 2 on a box with one GPU and no sharing switch: refused before any rank starts
    env.pop("OSD_BENCH_SHARE_GPU")
    if torch.cuda.device_count() < 2:
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                             capture_output=True, text=True, timeout=300, cwd=root, env=env)
        assert out.returncode == 2 and "GPU(s) visible" in out.stderr


def test_config5_multiscale_five_shot_training_steps():
    """BASELINE.json configs[4] at full size: bs=4, S=5 queries per image (20 x 127x127), the target's short edge cycling
    through {640, 800, 1024} (640x832, 800x1024, 1024x1312) from step to step on ONE engine (buffers keyed by shape).
    Properties: finite losses, positives found, every trainable bucket updated, and a repeat of the first shape after the
    cycle still runs (no stale per-shape state); bf16 like configs[2]."""
    from oneshotdet_amd import train
    np_sd = synth.make_state_dict(spec.hot_path_shapes())
    eng = train.TrainEngine(np_sd, dtype=torch.bfloat16, lr=0.002)
    B, S = 4, 5
    q = torch.from_numpy(synth.make_images("c5.query", B * S, 127, 127, seed=3)).cuda()
    sizes = [(640, 832), (800, 1024), (1024, 1312), (640, 832)]
    w0 = eng.flat_w.clone()
    for step, (h, w) in enumerate(sizes):
        img = torch.from_numpy(synth.make_images("c5.target.%d" % step, B, h, w, seed=step)).cuda()
        gts = synth.make_gt_boxes(B, h, w, seed=20 + step, max_boxes=6)
        G = max(len(g) for g in gts)
        gtb = torch.zeros(B, G, 4)
        for i, g in enumerate(gts):
            gtb[i, :len(g)] = torch.from_numpy(g)
        cnt = torch.tensor([len(g) for g in gts], dtype=torch.int32)
        losses = eng.train_step(img, q, gtb.cuda(), cnt.cuda()).cpu()
        assert torch.isfinite(losses).all(), (step, losses)
        assert losses[3] > 0 and losses[:3].sum() > 0
    torch.cuda.synchronize()
    moved = (eng.flat_w != w0)
    for name, (lo, hi) in eng.exchange.ranges.items():
        assert moved[lo:hi].any(), name
    assert moved.float().mean() > 0.5
    assert torch.isfinite(eng.flat_w).all()


def test_image_without_ground_truth_and_empty_second_stage_inputs():
    """Edge cases the synthetic benchmark never hits: a training batch in which one image has NO ground-truth box (all of
    its locations are negatives: finite losses and gradients, the positives of the other image still counted), and the
    second stage on zero valid proposals for one image (zero detections for it, the other image unaffected)."""
    from oneshotdet_amd import model, train
    B, H, W = 2, 128, 160
    np_sd = synth.make_state_dict(spec.full_model_shapes())
    hot = {k: v for k, v in np_sd.items() if k in spec.hot_path_shapes()}
    eng = train.TrainEngine(hot, dtype=torch.float32)
    img = torch.from_numpy(synth.make_images("e.img", B, H, W, seed=4)).cuda()
    q = torch.from_numpy(synth.make_images("e.q", B, 63, 63, seed=4)).cuda()
    gtb = torch.zeros(B, 2, 4)
    gtb[0, 0] = torch.tensor([20.0, 30.0, 90.0, 100.0])
    cnt = torch.tensor([1, 0], dtype=torch.int32)
    losses = eng.forward_backward(img, q, gtb.cuda(), cnt.cuda()).cpu()
    assert torch.isfinite(losses).all() and losses[3] > 0
    assert torch.isfinite(eng.flat_g).all() and float(eng.flat_g.abs().sum()) > 0
    torch.cuda.synchronize()
    pb, ps, pc = eng.proposals
    assert int(pc[1]) > 0 and float(ps[1, int(pc[1]) - 1]) < 1.0          # no ground-truth row appended for image 1
    assert float(ps[0, int(pc[0]) - 1]) == 1.0                            # image 0 ends with its ground-truth box
    # the same image alone gives the same number of positives
    one = eng.forward_backward(img[:1], q[:1], gtb[:1].cuda(), cnt[:1].cuda()).cpu()
    assert int(one[3]) == int(losses[3])
    inf = model.HotPathEngine(np_sd, dtype=torch.float32)
    out = inf.detect(img, q, cuda_nms=False)
    boxes, counts = out["proposals"][0], out["proposals"][2].clone()
    counts[1] = 0
    det = inf.box_detect(out["features"], out["query_features"], (63, 63), boxes, counts, H, W, cuda_nms=False)
    ref = inf.box_detect(out["features"], out["query_features"], (63, 63), boxes, out["proposals"][2], H, W, cuda_nms=False)
    assert int(det["counts"][1]) == 0 and int(det["counts"][0]) == int(ref["counts"][0]) > 0
    k = int(ref["counts"][0])
    assert torch.equal(det["boxes"][0, :k], ref["boxes"][0, :k])


def test_engine_with_ordered_weight_gradients_matches_default_engine():
    """TrainEngine(ordered_wgrad=True): every conv_wgrad launch of the step goes through the stored-partials + ordered
    reduction path (its own tuner cache entries, 1 GiB of scratch per weight-gradient stream): same losses, the same
    gradients to rounding and the same weights after two steps as the default (atomic) engine."""
    from oneshotdet_amd import ops, train
    outs, keep = {}, None
    try:
        for ordered in (False, True):
            eng, img, q, gtb, cnt = _engine_and_inputs("f32", "small")
            if ordered:
                eng = train.TrainEngine(synth.make_state_dict(spec.hot_path_shapes()), dtype=torch.float32, ordered_wgrad=True)
                assert eng.ordered_wgrad and len(ops._WGRAD_WS) == 2
                keep = eng
            with ops.tuning():
                l0 = eng.forward_backward(img, q, gtb, cnt).clone()
            g0 = eng.flat_g.clone()
            eng.optimizer_step()
            eng.train_step(img, q, gtb, cnt)
            eng.join()
            torch.cuda.synchronize()
            outs[ordered] = (l0.cpu(), g0.cpu(), eng.flat_w.clone().cpu())
    finally:
        keep.close()
    assert not ops._WGRAD_WS            # close() released both scratch buffers
    torch.testing.assert_close(outs[True][0], outs[False][0], rtol=1e-6, atol=0)
    assert (outs[True][1] - outs[False][1]).abs().max() <= 1e-4 * outs[False][1].abs().max()
    # weights after a second step: rounding-level gradient differences (the other kernels' atomics) pass through one update
    assert float((outs[True][2] - outs[False][2]).norm() / outs[False][2].norm()) < 1e-5


def test_ordered_weight_gradients_cover_the_second_stage_stream():
    """TrainEngine(second_stage=True, ordered_wgrad=True): the box head's weight gradients run on the proposal stream, which
    now has its own scratch buffer — two runs of the same step (same sampler keys) give BIT-IDENTICAL gradients for every
    roi_heads.box conv / linear weight.  The backbone gradients also receive the second stage's ROI-pool backward,
    which scatters with fp32 atomics: equal to rounding only (the engine warns about exactly that)."""
    import warnings
    from oneshotdet_amd import ops, train
    B, H, W, S, qh, qw = gu.CASES["small"]
    img, q = gu.case_inputs("small")
    gts = synth.make_gt_boxes(B, H, W, seed=3, max_boxes=3)
    gtb = torch.zeros(B, max(len(g) for g in gts), 4)
    for i, g in enumerate(gts):
        gtb[i, :len(g)] = torch.from_numpy(g)
    cnt = torch.tensor([len(g) for g in gts], dtype=torch.int32).cuda()
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        eng = train.TrainEngine(synth.make_state_dict(spec.full_model_shapes()), dtype=torch.float32, second_stage=True,
                                ordered_wgrad=True)
        assert any("ROI-pool backward" in str(x.message) for x in w)
    try:
        assert len(ops._WGRAD_WS) == 3 and eng.pstream.cuda_stream in ops._WGRAD_WS
        eng.box_keys = torch.rand((B, spec.POST_NMS_TOP_N_TRAIN + gtb.shape[1]), device="cuda",
                                  generator=torch.Generator(device="cuda").manual_seed(5))
        runs = []
        for _ in range(2):
            eng.forward_backward(torch.from_numpy(img).cuda(), torch.from_numpy(q).cuda(), gtb.cuda(), cnt)
            torch.cuda.synchronize()
            runs.append({k: v.clone() for k, v in eng.named_grads().items()})
        for k in runs[0]:
            # (the FCOS head's gradients pass through the loss kernel's atomically summed normalisers: last-bit differences)
            if k.startswith("roi_heads.box.") and k.endswith(".weight") and runs[0][k].dim() >= 2:
                assert torch.equal(runs[0][k], runs[1][k]), k
        for k in runs[0]:
            a, b = runs[0][k].float(), runs[1][k].float()
            assert float((a - b).abs().max()) <= 1e-4 * float(b.abs().max()) + 1e-12, k
    finally:
        eng.close()
    assert not ops._WGRAD_WS


def test_captured_training_step_matches_eager_steps():
    """TrainEngine.capture / replay_step (hipGraph replay of forward + loss + backward, then of the optimiser; bench.py
    --graph): three replayed steps on changing data give the losses and the weights of three eager steps (fp32; atomics
    order aside).  capture() must work on an engine that has already run deferred-join steps with the proposal-depth
    feedback armed — both are switched off for the captured graph."""
    from oneshotdet_amd import train
    np_sd = synth.make_state_dict(spec.hot_path_shapes())
    B, h, w = 2, 128, 160
    q = torch.from_numpy(synth.make_images("cg.q", B, 63, 63, seed=1)).cuda()
    batches = []
    for step in range(4):
        img = torch.from_numpy(synth.make_images("cg.img.%d" % step, B, h, w, seed=step)).cuda()
        gts = synth.make_gt_boxes(B, h, w, seed=60 + step, max_boxes=3)
        gtb = torch.zeros(B, 3, 4)
        for i, g in enumerate(gts):
            gtb[i, :len(g)] = torch.from_numpy(g)
        batches.append((img, gtb.cuda(), torch.tensor([len(g) for g in gts], dtype=torch.int32).cuda()))
    out = {}
    for graphed in (False, True):
        eng = train.TrainEngine(np_sd, dtype=torch.float32, lr=0.002)
        eng.defer_join = True
        first = eng.train_step(batches[0][0], q, batches[0][1], batches[0][2]).clone()
        if graphed:
            eng.join()
            w0, m0, n0 = eng.flat_w.clone(), eng._sgd["buf"].clone(), eng._sgd["steps"]
            eng.capture(batches[0][0], q, batches[0][1], batches[0][2], warmup=1)      # runs training steps itself:
            eng.flat_w.copy_(w0)                                                            # restore the state after step 1
            eng._sgd["buf"].copy_(m0)
            eng._sgd["steps"] = n0
            eng.repack()
            losses = [eng.replay_step(img, q, gtb, cnt).clone() for img, gtb, cnt in batches[1:]]
        else:
            losses = [eng.train_step(img, q, gtb, cnt).clone() for img, gtb, cnt in batches[1:]]
        eng.join()
        torch.cuda.synchronize()
        out[graphed] = (torch.stack([first] + losses).cpu(), eng.flat_w.clone().cpu())
    # run-to-run spread of ONE mode is ~5e-5 on the losses (atomics); weights by relative L2, as the deferred-join test
    np.testing.assert_allclose(out[True][0].numpy(), out[False][0].numpy(), rtol=2e-3, atol=1e-6)
    assert float((out[True][1] - out[False][1]).norm() / out[False][1].norm()) < 1e-4


def test_replay_then_eager_backward_then_replay_does_not_double_count_gradients():
    """the per-step zeroing of the flat gradient buffer is a host flag (`_grads_clean`) kept in step with the fused
    update that zeroes what it has read.  A captured forward + backward must carry its own zeroing: the sequence replay_step ->
    EAGER forward_backward (gradients left in the buffer, no update) -> replay_step has to give the weights of replay_step ->
    replay_step on the same data, and gradients written into flat_g by hand before a replay must not be applied."""
    from oneshotdet_amd import train
    np_sd = synth.make_state_dict(spec.hot_path_shapes())
    B, h, w = 2, 128, 160
    q = torch.from_numpy(synth.make_images("rg.q", B, 63, 63, seed=1)).cuda()
    batches = []
    for step in range(3):
        img = torch.from_numpy(synth.make_images("rg.img.%d" % step, B, h, w, seed=step)).cuda()
        gts = synth.make_gt_boxes(B, h, w, seed=80 + step, max_boxes=3)
        gtb = torch.zeros(B, 3, 4)
        for i, g in enumerate(gts):
            gtb[i, :len(g)] = torch.from_numpy(g)
        batches.append((img, gtb.cuda(), torch.tensor([len(g) for g in gts], dtype=torch.int32).cuda()))
    out = {}
    for mixed in (False, True):
        eng = train.TrainEngine(np_sd, dtype=torch.float32, lr=0.002)
        try:
            eng.capture(batches[0][0], q, batches[0][1], batches[0][2], warmup=1)
            w0 = None
            eng.replay_step(batches[1][0], q, batches[1][1], batches[1][2])
            if mixed:
                eng.forward_backward(batches[0][0], q, batches[0][1], batches[0][2])      # gradients stay in flat_g, no update
                torch.cuda.synchronize()
                assert float(eng.flat_g.abs().max()) > 0
            eng.replay_step(batches[2][0], q, batches[2][1], batches[2][2])
            if mixed:
                torch.cuda.synchronize()
                w0 = eng.flat_w.clone()
                eng.flat_g.fill_(1e3)                                                       # a hand-written gradient before a replay
                eng.replay_step(batches[2][0], q, batches[2][1], batches[2][2])
            else:
                eng.replay_step(batches[2][0], q, batches[2][1], batches[2][2])
            torch.cuda.synchronize()
            out[mixed] = eng.flat_w.clone().cpu()
            if w0 is not None:      # an update from a 1e3-everywhere gradient would move every weight by >= lr * 1e3 = 2
                assert float((eng.flat_w - w0).abs().max()) < 0.5
        finally:
            eng.close()
    assert float((out[True] - out[False]).norm() / out[False].norm()) < 1e-4


def test_deferred_join_matches_joined_steps():
    """train_step(defer_join=True) leaves the step's tail (last weight gradients, update, repack, proposals) on the side
    streams and lets the next step's frozen layers run beside it.  Same data, same steps: the losses of every step and the
    weights after four steps equal those of the joined engine (fp32; the atomics' order is the only difference), the
    state_dict() / join() accessors see finished work, and a changing input shape between steps is handled."""
    from oneshotdet_amd import train
    np_sd = synth.make_state_dict(spec.hot_path_shapes())
    B = 2
    q = torch.from_numpy(synth.make_images("dj.q", B, 63, 63, seed=1)).cuda()
    sizes = [(128, 160), (128, 160), (96, 128), (128, 160)]
    batches = []
    for step, (h, w) in enumerate(sizes):
        img = torch.from_numpy(synth.make_images("dj.img.%d" % step, B, h, w, seed=step)).cuda()
        gts = synth.make_gt_boxes(B, h, w, seed=40 + step, max_boxes=3)
        G = max(len(g) for g in gts)
        gtb = torch.zeros(B, G, 4)
        for i, g in enumerate(gts):
            gtb[i, :len(g)] = torch.from_numpy(g)
        batches.append((img, gtb.cuda(), torch.tensor([len(g) for g in gts], dtype=torch.int32).cuda()))
    results = {}
    for mode in (False, True):
        eng = train.TrainEngine(np_sd, dtype=torch.float32, lr=0.002)
        eng.defer_join = mode
        losses = [eng.train_step(img, q, gtb, cnt) for img, gtb, cnt in batches]
        if mode:
            assert eng._deferred is not None              # the last step's tail is still un-joined on this stream
        sd = eng.state_dict()                              # joins
        assert eng._deferred is None
        torch.cuda.synchronize()
        results[mode] = (torch.stack(losses).cpu(), eng.flat_w.clone().cpu(), sd["rpn.head.cls_tower.0.weight"].cpu())
        pb, ps, pc = eng.proposals
        assert int(pc.min()) > 0
    la, wa, _ = results[False]
    lb, wb, ta = results[True]
    np.testing.assert_allclose(lb.numpy(), la.numpy(), rtol=2e-3, atol=1e-6)      # run-to-run spread of one mode: ~5e-5
    assert float((wb - wa).norm() / wa.norm()) < 1e-4
    assert torch.isfinite(ta).all()


@pytest.mark.parametrize("name", ["small", "shots5"])
def test_bf16_training_step_against_the_bf16_emulating_oracle(name):
    """Second tier for the MEASURED dtype (bf16 training step): losses and parameter gradients against the oracle's
    reduced-precision mode (oracle.hotpath_ref.Emulation: forward AND backward rounded where the engine stores; the frozen
    layer1.0 fuses conv3 + downsample), free running.  The floor under this comparison is the rounded pipeline's own
    sensitivity (tests/test_oracle_emulation.py: a 1e-6 weight perturbation moves its gradients by 0.16 median / 0.25 worst
    relative L2 — as far as bf16 is from fp32), so the bars are the first tier's: relative L2 <= 0.35, cosine >= 0.96 per
    tensor, plus a median over the tensors <= 0.25 and losses within 2e-2.  The tight statement is
    tests/test_gpu_launch_replay.py: every launch of this step within one bf16 ulp of its restatement."""
    B, H, W, S, qh, qw = gu.CASES[name]
    eng, img, q, gtb, cnt = _engine_and_inputs("bf16", name)
    losses = eng.forward_backward(img, q, gtb, cnt).cpu().numpy()
    grads = eng.named_grads()
    sd = {k: v.clone().requires_grad_(not spec.is_frozen(k)) for k, v in
          orc.to_torch_state_dict(synth.make_state_dict(spec.hot_path_shapes())).items()}
    o = orc.hot_path_forward(img.cpu(), q.cpu(), sd, shots=S, emu=orc.Emulation(torch.bfloat16, fused_downsample=("layer1",)))
    gts = synth.make_gt_boxes(B, H, W, seed=3, max_boxes=3)
    c, r, t, info = orc.fcos_loss(o["logits"], o["bbox_reg"], o["centerness"], gts, focal="cuda")
    (c + r + t).backward()
    assert int(losses[3]) == info["num_pos"]
    np.testing.assert_allclose(losses[:3], [c.item(), r.item(), t.item()], rtol=2e-2)
    l2s = []
    for k, v in sd.items():
        if v.grad is None or k.endswith(".scale") or float(v.grad.abs().max()) == 0.0:
            continue
        g, ref = grads[k].float().cpu().reshape(-1), v.grad.reshape(-1)
        l2 = float((g - ref).norm() / ref.norm())
        cos = float((g * ref).sum() / (g.norm() * ref.norm()))
        assert l2 <= 0.35 and cos >= 0.96, (k, l2, cos)
        l2s.append(l2)
    l2s.sort()
    print("\n%s bf16 step vs bf16-emulating oracle: losses %s vs %s; gradient relative L2 median %.3f worst %.3f over %d tensors"
          % (name, losses[:3], [c.item(), r.item(), t.item()], l2s[len(l2s) // 2], l2s[-1], len(l2s)))
    assert l2s[len(l2s) // 2] <= 0.25
