"""GPU (-m gpu): the one-pass GroupNorm + ReLU kernels (csrc/groupnorm_onepass.hip; fcos.py:29-37's GroupNorm + ReLU of a tower
layer, forward and backward) against the two-launch kernels they replace and against torch autograd.

The two forms compute the same formulas and differ only in the ORDER the fp32 partial sums of the statistics are added (64 slabs
per image before, one partial per 112- / 256-pixel workgroup now), so the bars are: saved statistics (ab) and d gamma / d beta to
fp32 summation noise (rtol 2e-5 of the tensor's absmax), bf16 outputs within one bf16 rounding step of each other, and — because
the new kernels add their partials in a fixed order — BIT-IDENTICAL results from launch to launch, beside a busy stream, and on two
streams at once.  The inter-workgroup hand-off is what these tests are for: uneven load, repeated launches on the same sync
words, jobs of 1 - 115 workgroups."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _levels(n, c, sizes, seed=0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    xs = [(torch.randn((n, h, w, c), device="cuda", generator=g) * 2 + 0.3).bfloat16() for h, w in sizes]
    dts = [(torch.randn((n, h, w, c), device="cuda", generator=g) * (10.0 if i == 1 else 1.0)).bfloat16() for i, (h, w) in enumerate(sizes)]
    gamma = torch.randn(c, device="cuda", generator=g) * 0.7
    gamma[::7] = 0.0
    gamma[2::13] = -0.4
    beta = torch.randn(c, device="cuda", generator=g)
    return xs, dts, gamma, beta


def _run(ops, xs, dts, gamma, beta, groups, onepass):
    old = (ops.GN_ONEPASS, ops.GN_ONEPASS_FWD, ops.GN_ONEPASS_BWD)
    ops.GN_ONEPASS = ops.GN_ONEPASS_FWD = ops.GN_ONEPASS_BWD = onepass
    try:
        c = xs[0].shape[-1]
        ys, ab = ops.groupnorm_relu_levels(xs, gamma, beta, groups, 1e-5)
        dg, db = torch.zeros(c, device="cuda"), torch.zeros(c, device="cuda")
        dus = ops.groupnorm_relu_bwd_levels(xs, dts, ab, gamma, beta, dg, db, groups)
    finally:
        ops.GN_ONEPASS, ops.GN_ONEPASS_FWD, ops.GN_ONEPASS_BWD = old
    return ys, ab, dus, dg, db


def _close_bf16(a, b, what):
    """within one bf16 rounding step of each other (plus the fp32 noise of a u + b where the two terms cancel), and exactly equal
    almost everywhere"""
    a, b = a.float(), b.float()
    d = (a - b).abs()
    bound = torch.maximum(a.abs(), b.abs()) * 2.0 ** -7 + 2e-5 * float(b.abs().max())
    assert bool((d <= bound).all()), (what, float((d / bound).max()))
    assert float((d > 0).float().mean()) < 0.02, (what, float((d > 0).float().mean()))


@pytest.mark.parametrize("case", [
    (2, 256, 32, [(100, 128), (50, 64), (25, 32), (13, 16), (7, 8)]),      # the FCOS tower at 800 x 1024 (P3 .. P7)
    (3, 256, 32, [(13, 17), (1, 1), (3, 37), (16, 16), (1, 113)]),          # ragged: partial workgroups, one-pixel maps, whole tiles
    (2, 128, 16, [(40, 52), (5, 7)]),                                       # 16 chunks per pixel: 32 pixel lanes
    (2, 512, 32, [(20, 26), (9, 3)]),                                       # two 16-byte chunks per group
    (1, 64, 8, [(33, 41)]),
])
def test_onepass_groupnorm_equals_the_two_launch_kernels(case):
    from oneshotdet_amd import ops
    n, c, groups, sizes = case
    xs, dts, gamma, beta = _levels(n, c, sizes)
    assert ops.gn_onepass_ok(xs[0], groups)
    y1, ab1, du1, dg1, db1 = _run(ops, xs, dts, gamma, beta, groups, True)
    y2, ab2, du2, dg2, db2 = _run(ops, xs, dts, gamma, beta, groups, False)
    torch.testing.assert_close(ab1, ab2, rtol=2e-5, atol=2e-5 * float(ab2.abs().max()))
    for l, (a, b) in enumerate(zip(y1, y2)):
        _close_bf16(a, b, "y level %d" % l)
    # the backward kernels read the SAME saved statistics here? no: each form read its own ab (they differ by fp32 noise), so the
    # masks can flip on elements with |z| ~ 1e-6 |a u|: compare through the bf16 bar on all but a handful of elements
    for l, (a, b) in enumerate(zip(du1, du2)):
        a, b = a.float(), b.float()
        d = (a - b).abs()
        bound = torch.maximum(a.abs(), b.abs()) * 2.0 ** -6 + 2e-3 * float(b.abs().max())
        assert float((d > bound).float().mean()) < 1e-4, ("du level %d" % l, float((d > bound).float().mean()))
    torch.testing.assert_close(dg1, dg2, rtol=1e-3, atol=1e-3 * float(dg2.abs().max()))
    torch.testing.assert_close(db1, db2, rtol=1e-3, atol=1e-3 * float(db2.abs().max()))
    assert ops.gn_onepass_errors() == 0


def test_onepass_groupnorm_backward_on_the_same_statistics_is_the_two_launch_backward():
    """Given the SAME saved statistics (ab) the two backward forms evaluate the same expression per element; only the group sums
    (c1, c2) come from another summation order: du within one bf16 step everywhere, d gamma / d beta to fp32 noise."""
    from oneshotdet_amd import ops
    n, c, groups, sizes = 2, 256, 32, [(100, 128), (50, 64), (25, 32), (13, 16), (7, 8)]
    xs, dts, gamma, beta = _levels(n, c, sizes, seed=5)
    old = (ops.GN_ONEPASS, ops.GN_ONEPASS_FWD, ops.GN_ONEPASS_BWD)
    try:
        ops.GN_ONEPASS = ops.GN_ONEPASS_FWD = ops.GN_ONEPASS_BWD = False
        _, ab = ops.groupnorm_relu_levels(xs, gamma, beta, groups, 1e-5)
        dg2, db2 = torch.zeros(c, device="cuda"), torch.zeros(c, device="cuda")
        du2 = ops.groupnorm_relu_bwd_levels(xs, dts, ab, gamma, beta, dg2, db2, groups)
        ops.GN_ONEPASS = ops.GN_ONEPASS_BWD = True
        dg1, db1 = torch.zeros(c, device="cuda"), torch.zeros(c, device="cuda")
        du1 = ops.groupnorm_relu_bwd_levels(xs, dts, ab, gamma, beta, dg1, db1, groups)
    finally:
        ops.GN_ONEPASS, ops.GN_ONEPASS_FWD, ops.GN_ONEPASS_BWD = old
    for l, (a, b) in enumerate(zip(du1, du2)):
        a, b = a.float(), b.float()
        d = (a - b).abs()
        bound = torch.maximum(a.abs(), b.abs()) * 2.0 ** -7 + 1e-5 * float(b.abs().max())
        assert bool((d <= bound).all()), ("du level %d" % l, float((d / bound).max()))
    torch.testing.assert_close(dg1, dg2, rtol=2e-5, atol=2e-5 * float(dg2.abs().max()))
    torch.testing.assert_close(db1, db2, rtol=2e-5, atol=2e-5 * float(db2.abs().max()))
    assert ops.gn_onepass_errors() == 0


def test_onepass_groupnorm_matches_autograd():
    from oneshotdet_amd import ops
    n, c, groups = 2, 256, 32
    sizes = [(50, 64), (13, 16), (7, 8)]
    xs, dts, gamma, beta = _levels(n, c, sizes, seed=2)
    y1, ab1, du1, dg1, db1 = _run(ops, xs, dts, gamma, beta, groups, True)
    g = gamma.cpu().clone().requires_grad_(True)
    b = beta.cpu().clone().requires_grad_(True)
    for x, dt, y, du in zip(xs, dts, y1, du1):
        xr = x.float().cpu().permute(0, 3, 1, 2).clone().requires_grad_(True)
        ref = F.relu(F.group_norm(xr, groups, g, b, eps=1e-5))
        (ref * dt.float().cpu().permute(0, 3, 1, 2)).sum().backward()
        assert (y.float().cpu().permute(0, 3, 1, 2) - ref.detach()).abs().max() <= 1e-2 * ref.abs().max()
        assert (du.float().cpu().permute(0, 3, 1, 2) - xr.grad).abs().max() <= 2e-2 * xr.grad.abs().max()
    assert (dg1.cpu() - g.grad).abs().max() <= 2e-3 * g.grad.abs().max()
    assert (db1.cpu() - b.grad).abs().max() <= 2e-3 * b.grad.abs().max()


def test_onepass_groupnorm_is_deterministic_beside_a_busy_stream_and_on_two_streams():
    """The hand-off under uneven load: 40 launches back to back on the same sync words while another stream keeps the chip busy with
    matrix products (workgroups of a job start far apart, the consumer's CU has read the same lines a launch earlier), then two
    streams running the kernels at the same time on sync buffers of their own — every output bit-identical to the quiet run."""
    from oneshotdet_amd import ops
    n, c, groups = 4, 256, 32
    sizes = [(100, 128), (50, 64), (25, 32), (13, 16), (7, 8)]
    xs, dts, gamma, beta = _levels(n, c, sizes, seed=3)
    y0, ab0, du0, _, _ = _run(ops, xs, dts, gamma, beta, groups, True)
    torch.cuda.synchronize()
    a = torch.randn((4096, 4096), device="cuda").bfloat16()
    side, side2 = torch.cuda.Stream(), torch.cuda.Stream()
    for rep in range(40):
        if rep % 2 == 0:
            with torch.cuda.stream(side):
                for _ in range(3):
                    a @ a
        y, ab, du, _, _ = _run(ops, xs, dts, gamma, beta, groups, True)
        assert all(torch.equal(p, q) for p, q in zip(y, y0)), rep
        assert torch.equal(ab, ab0), rep
        assert all(torch.equal(p, q) for p, q in zip(du, du0)), rep
    torch.cuda.synchronize()
    outs = {}
    for s in (side, side2):
        s.wait_stream(torch.cuda.current_stream())
    for rep in range(10):
        for s in (side, side2):
            with torch.cuda.stream(s):
                outs[s] = _run(ops, xs, dts, gamma, beta, groups, True)
        torch.cuda.synchronize()
        for s in (side, side2):
            y, ab, du, _, _ = outs[s]
            assert all(torch.equal(p, q) for p, q in zip(y, y0)), rep
            assert all(torch.equal(p, q) for p, q in zip(du, du0)), rep
    assert ops.gn_onepass_errors() == 0


def test_onepass_groupnorm_hands_maps_too_large_for_one_resident_job_to_the_two_launch_kernels():
    """A (level, image) whose pixels need more workgroups than can be resident together (> 192 of 320 / 128 pixels) must not run
    the hand-off at all — the spin would end in the error word: the C entry refuses it (OSD_ERR_UNSUPPORTED) and ops falls back to
    the two-launch kernels, bit for bit."""
    from oneshotdet_amd import _lib, ops
    xs, dts, gamma, beta = _levels(1, 256, [(400, 512)], seed=9)       # 204,800 pixels: 640 forward / 1,600 backward workgroups
    y1, ab1, du1, dg1, db1 = _run(ops, xs, dts, gamma, beta, 32, True)
    y2, ab2, du2, dg2, db2 = _run(ops, xs, dts, gamma, beta, 32, False)
    assert torch.equal(y1[0], y2[0]) and torch.equal(ab1, ab2) and torch.equal(du1[0], du2[0])
    with pytest.raises(_lib.OsdError):
        import ctypes as C
        hws = (C.c_int32 * 1)(400 * 512)
        sync = torch.zeros(4096, device="cuda", dtype=torch.int32)
        ws = torch.empty(1 << 22, device="cuda")
        y = torch.empty_like(xs[0])
        ab = torch.empty((1, 4, 1, 256), device="cuda")
        _lib.call("osd_groupnorm_relu_fwd_levels_onepass", 1, ops._ptr_array(xs), ops._ptr_array([y]), hws, ops._ptr(gamma), ops._ptr(beta),
                  ops._ptr(ab), ops._ptr(ws), ops._ptr(sync), 1, 256, 32, 1e-5, ops._dt(xs[0]), ops._stream())
    assert ops.gn_onepass_errors() == 0


def test_onepass_groupnorm_timeout_is_loud():
    """VERDICT r5 item 4 / ADVICE r5: a workgroup that gives up waiting for its job must not "go on with what it has" silently.  The
    diagnostic entry launches the forward kernel WITHOUT its last workgroup and with a short spin: the incomplete job's workgroups
    time out, set the error word, and write NaN into every output element and into the saved statistics of that (level, image);
    complete jobs are untouched; `ops.gn_onepass_check()` (what bench.py and TrainEngine.state_dict call) raises."""
    import ctypes as C
    from oneshotdet_amd import _lib, ops
    n, c, groups = 2, 256, 32
    sizes = [(50, 64), (25, 32)]                        # 3200 / 800 pixels per image: 10 / 3 workgroups of 320 pixels each
    xs, _, gamma, beta = _levels(n, c, sizes, seed=11)
    ys = [torch.zeros_like(x) for x in xs]
    ab = torch.zeros((2, 4, n, c), device="cuda")
    hws = (C.c_int32 * 2)(*[h * w for h, w in sizes])
    sync = torch.zeros(32 * 64, device="cuda", dtype=torch.int32)          # a buffer of its own: left dirty by the self-test
    ws = torch.empty(int(_lib.load().osd_groupnorm_onepass_workspace_bytes(2, hws, n, c, groups, 0)) // 4, device="cuda")
    assert ops.gn_onepass_errors() == 0
    _lib.call("osd_groupnorm_onepass_selftest_timeout", 2, ops._ptr_array(xs), ops._ptr_array(ys), hws, ops._ptr(gamma), ops._ptr(beta),
              ops._ptr(ab), ops._ptr(ws), ops._ptr(sync), n, c, groups, 1e-5, ops._dt(xs[0]), 2000, ops._stream())
    torch.cuda.synchronize()
    assert int(sync[2].item()) == 1
    # the dropped workgroup is the LAST ticket: level 1, image 1, last part.  Its job's other workgroups wrote NaN ...
    last = ys[1][1].float().reshape(-1, c)
    px = 320
    assert torch.isnan(last[:2 * px]).all()
    assert (last[2 * px:] == 0).all()                   # ... the dropped one wrote nothing
    assert torch.isnan(ab[1, :, 1]).all()
    # ... and every complete job is what the normal launch gives
    y_ref, ab_ref = ops.groupnorm_relu_levels(xs, gamma, beta, groups=groups)
    assert torch.equal(ys[0], y_ref[0]) and torch.equal(ys[1][0], y_ref[1][0])
    assert torch.equal(ab[0], ab_ref[0]) and torch.equal(ab[1, :, 0], ab_ref[1, :, 0])
    # the host-side check: the error word of a REGISTERED sync buffer raises where the host synchronises
    key = ("selftest", 0)
    ops._GN1P_SYNC[key] = sync
    try:
        with pytest.raises(_lib.OsdError):
            ops.gn_onepass_check("test")
    finally:
        del ops._GN1P_SYNC[key]
    assert ops.gn_onepass_errors() == 0
