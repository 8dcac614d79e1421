"""GPU (-m gpu): NMS against the reference's own known-answer vectors, the proposal pipeline against the oracle,
and size-independent properties at full size."""
import numpy as np
import pytest
import torch

import golden_utils as gu
from oracle import hotpath_ref as orc

pytestmark = pytest.mark.gpu


def test_nms_reference_known_answers():
    """reference tests/test_nms.py:11-217 (recorded in tests/golden/nms_kat.npz); CPU rule (>=) as recorded."""
    from oneshotdet_amd import layers
    f = gu.load("nms_kat.npz")
    for i in range(int(f["n"])):
        keep = layers.nms(torch.from_numpy(f["boxes.%d" % i]).cuda(), torch.from_numpy(f["scores.%d" % i]).cuda(),
                          float(f["thresh.%d" % i]), cuda_semantics=False)
        np.testing.assert_array_equal(keep.cpu().numpy(), f["keep.%d" % i])


def test_nms_rules_and_edge_cases():
    from oneshotdet_amd import layers
    dev = "cuda"
    assert layers.nms(torch.zeros(0, 4, device=dev), torch.zeros(0, device=dev), 0.5).shape == (0,)
    b = torch.tensor([[0, 0, 9, 9], [0, 0, 9, 9], [20, 20, 30, 30]], dtype=torch.float32, device=dev)
    s = torch.tensor([0.5, 0.9, 0.1], device=dev)
    assert layers.nms(b, s, 1.0, cuda_semantics=False).tolist() == [1, 2]     # nms_cpu.cpp:60  '>='
    assert layers.nms(b, s, 1.0, cuda_semantics=True).tolist() == [0, 1, 2]   # nms.cu:60       '>'
    with pytest.raises(RuntimeError):
        layers.nms(b.cpu(), s.cpu(), 0.5)


@pytest.mark.parametrize("n", [1, 63, 64, 65, 1000, 4097])
def test_nms_random_vs_oracle(n):
    from oneshotdet_amd import layers
    rng = np.random.RandomState(n)
    xy = rng.uniform(0, 200, (n, 2)).astype(np.float32)
    wh = rng.uniform(5, 80, (n, 2)).astype(np.float32)
    boxes = np.concatenate([xy, xy + wh], 1)
    scores = rng.permutation(n).astype(np.float32) / n          # distinct scores: no tie ambiguity
    for thr, cuda in ((0.5, True), (0.8, False), (0.3, True)):
        ref = orc.nms(boxes, scores, thr, cuda_semantics=cuda)
        got = layers.nms(torch.from_numpy(boxes).cuda(), torch.from_numpy(scores).cuda(), thr, cuda_semantics=cuda)
        np.testing.assert_array_equal(got.cpu().numpy(), ref)
    # idempotence: NMS of the survivors keeps all of them
    keep = layers.nms(torch.from_numpy(boxes).cuda(), torch.from_numpy(scores).cuda(), 0.5)
    again = layers.nms(torch.from_numpy(boxes).cuda()[keep], torch.from_numpy(scores).cuda()[keep], 0.5)
    assert again.numel() == keep.numel()


def test_nms_iou_rounding_matches_the_cpu_kernel():
    """The threshold is set to the fp32 IoU of one pair, computed like nms_cpu.cpp:40-60 (separately rounded mul / add / sub /
    div): under the CPU rule '>=' that pair is suppressed only if the kernel rounds its IoU identically (an FMA-contracted
    union differs in the last bit for about one pair in three)."""
    from oneshotdet_amd import layers
    rng = np.random.RandomState(7)
    n = 384
    ctr = rng.uniform(40, 160, (n, 2)).astype(np.float32)
    wh = rng.uniform(30, 90, (n, 2)).astype(np.float32)
    boxes = np.concatenate([ctr - wh / 2, ctr + wh / 2], 1).astype(np.float32)
    scores = rng.permutation(n).astype(np.float32) / n
    one = np.float32(1)
    area = (boxes[:, 2] - boxes[:, 0] + one) * (boxes[:, 3] - boxes[:, 1] + one)
    b_dev, s_dev = torch.from_numpy(boxes).cuda(), torch.from_numpy(scores).cuda()
    tried = 0
    for _ in range(400):
        i, j = rng.randint(0, n, 2)
        w = np.maximum(np.float32(0), np.minimum(boxes[i, 2], boxes[j, 2]) - np.maximum(boxes[i, 0], boxes[j, 0]) + one)
        h = np.maximum(np.float32(0), np.minimum(boxes[i, 3], boxes[j, 3]) - np.maximum(boxes[i, 1], boxes[j, 1]) + one)
        inter = np.float32(w * h)
        thr = np.float32(inter / np.float32(np.float32(area[i] + area[j]) - inter))
        if i == j or not (0.2 < thr < 0.9):
            continue
        tried += 1
        for cuda in (False, True):
            ref = orc.nms(boxes, scores, float(thr), cuda_semantics=cuda)
            got = layers.nms(b_dev, s_dev, float(thr), cuda_semantics=cuda)
            np.testing.assert_array_equal(got.cpu().numpy(), ref, err_msg="pair (%d, %d) thr %r" % (i, j, thr))
        if tried == 48:
            break
    assert tried == 48


def test_nms_second_phase_when_first_candidates_do_not_suffice():
    """Heavy overlap among the best-scoring boxes: the first `limit` candidates yield fewer than max_keep survivors, so
    the device-side fallback (full mask + full scan) must produce the answer; an image of the same batch that needs no
    fallback must be unaffected."""
    from oneshotdet_amd import ops
    rng = np.random.RandomState(0)
    n, max_keep = 3000, 100
    centers = rng.uniform(50, 150, (8, 2)).astype(np.float32)
    boxes = np.zeros((2, n, 4), np.float32)
    for img in range(2):
        xy = rng.uniform(0, 900, (n, 2)).astype(np.float32)
        if img == 0:      # the 2000 best boxes are jittered copies of 8 clusters
            xy[:2000] = centers[rng.randint(0, 8, 2000)] + rng.uniform(-1, 1, (2000, 2)).astype(np.float32)
        boxes[img] = np.concatenate([xy, xy + 60], 1)
    scores = np.tile(np.linspace(1.0, 0.01, n, dtype=np.float32), (2, 1))      # already sorted, distinct
    bs, ss, idx, cnt = ops.rank_sort_gather(torch.from_numpy(scores).cuda(), torch.from_numpy(boxes).cuda(), n)
    ob, os_, op, oc = ops.nms_sorted(bs, ss, cnt, 0.5, max_keep, cuda_semantics=True)
    for img in range(2):
        ref = orc.nms(boxes[img], scores[img], 0.5, cuda_semantics=True)[:max_keep]
        k = int(oc[img])
        assert k == len(ref) == max_keep
        np.testing.assert_array_equal(op[img, :k].cpu().numpy(), ref)
        np.testing.assert_array_equal(ob[img, :k].cpu().numpy(), boxes[img][ref])


def _split_head(head, hw):
    logits, reg, ctr, off = [], [], [], 0
    for (h, w) in hw:
        blk = head[:, off:off + h * w].permute(0, 2, 1).reshape(head.shape[0], 6, h, w)
        logits.append(blk[:, 0:1]), reg.append(blk[:, 1:5]), ctr.append(blk[:, 5:6])
        off += h * w
    return logits, reg, ctr


@pytest.mark.parametrize("name", ["small", "nonsquare", "config1"])
def test_proposals_from_golden_head(name):
    """R10/R11: score, per-level top-k, decode, clip, NMS(0.8), top-2000 from the REFERENCE's head outputs; compared
    with the reference's proposals stored in the fixture (CPU rule)."""
    from oneshotdet_amd import model, spec
    B, H, W, S, qh, qw = gu.CASES[name]
    f = gu.load("case_%s.npz" % name)
    head = torch.from_numpy(f["head"])
    from oneshotdet_amd import ops
    sizes = spec.level_sizes(H, W) if (H % 32 == 0 and W % 32 == 0) else None
    if sizes is None:
        pytest.skip("non /32 size")
    logits, reg, ctr = _split_head(head, sizes)
    head_out = []
    for lg, rg, ct in zip(logits, reg, ctr):
        cc = torch.cat([lg, ct, torch.zeros_like(lg), torch.zeros_like(lg)], 1).permute(0, 2, 3, 1).contiguous().cuda()
        head_out.append((cc, rg.permute(0, 2, 3, 1).contiguous().cuda()))
    ob, os_, oc = model.run_proposals(head_out, H, W, spec.PRE_NMS_TOP_N_TEST, spec.POST_NMS_TOP_N_TEST,
                                      spec.NMS_THRESH, cuda_nms=False)
    for i in range(B):
        k = int(oc[i])
        rb, rs = f["proposals.%d.boxes" % i], f["proposals.%d.scores" % i]
        assert k == len(rb)
        sc = os_[i, :k].cpu().numpy()
        assert np.all(np.diff(sc) <= 0), "scores must be descending"
        assert gu.match_boxes(rb, rs, ob[i, :k].cpu().numpy(), sc) >= 0.999
        np.testing.assert_allclose(sc, rs, rtol=1e-5, atol=1e-7)


def test_append_gt_boxes_matches_reference_vectors():
    """osd_append_gt_boxes against the fixture recorded through the reference's add_gt_proposals (ragged proposal and
    ground-truth counts, an image with no ground truth), bit-exact; and TrainEngine's training proposals end with the
    ground-truth boxes at score 1."""
    from oneshotdet_amd import ops
    f = gu.load("add_gt.npz")
    n = int(f["n"])
    P = max(len(f["props.%d" % i]) for i in range(n)) + 2
    G = max(len(f["gt.%d" % i]) for i in range(n)) + 1
    boxes, scores, gt = torch.zeros(n, P, 4), torch.zeros(n, P), torch.zeros(n, G, 4)
    cnt, gcnt = torch.zeros(n, dtype=torch.int32), torch.zeros(n, dtype=torch.int32)
    for i in range(n):
        p, g = f["props.%d" % i], f["gt.%d" % i].reshape(-1, 4)
        boxes[i, :len(p)], scores[i, :len(p)], cnt[i] = torch.from_numpy(p), torch.from_numpy(f["scores.%d" % i]), len(p)
        gt[i, :len(g)], gcnt[i] = torch.from_numpy(g), len(g)
        boxes[i, len(p):] = 7.0            # stale rows past the count must not leak
    ob, os_, oc = ops.append_gt_boxes(boxes.cuda(), scores.cuda(), cnt.cuda(), gt.cuda(), gcnt.cuda())
    for i in range(n):
        k = int(oc[i])
        assert k == len(f["out_boxes.%d" % i])
        np.testing.assert_array_equal(ob[i, :k].cpu().numpy(), f["out_boxes.%d" % i])
        np.testing.assert_array_equal(os_[i, :k].cpu().numpy(), f["out_scores.%d" % i])
        assert float(ob[i, k:].abs().sum()) == 0 and float(os_[i, k:].abs().sum()) == 0


def test_training_proposals_end_with_ground_truth():
    from oneshotdet_amd import spec, synth as sy, train
    B, H, W = 2, 128, 160
    eng = train.TrainEngine(sy.make_state_dict(spec.hot_path_shapes()), dtype=torch.float32)
    img = torch.from_numpy(sy.make_images("t.img", B, H, W, seed=1)).cuda()
    q = torch.from_numpy(sy.make_images("t.q", B, 63, 63, seed=1)).cuda()
    gts = sy.make_gt_boxes(B, H, W, seed=9, max_boxes=3)
    G = max(len(g) for g in gts)
    gtb = torch.zeros(B, G, 4)
    for i, g in enumerate(gts):
        gtb[i, :len(g)] = torch.from_numpy(g)
    cnt = torch.tensor([len(g) for g in gts], dtype=torch.int32)
    eng.forward_backward(img, q, gtb.cuda(), cnt.cuda())
    torch.cuda.synchronize()
    pb, ps, pc = eng.proposals
    for i, g in enumerate(gts):
        k = int(pc[i])
        assert k > len(g)
        np.testing.assert_array_equal(pb[i, k - len(g):k].cpu().numpy(), g)
        assert bool((ps[i, k - len(g):k] == 1).all())
        s = ps[i, :k - len(g)]
        assert bool((s[:-1] >= s[1:]).all())


@pytest.mark.parametrize("case", ["levels_ties_drops", "single_segment", "fcos_sized", "big_fallback"])
def test_rank_sort_gather_against_a_plain_sort(case):
    """osd_rank_sort_gather against numpy: per-level top-n cut by (score desc, index asc), survivors of all levels in that
    order, boxes gathered.  Heavy ties (quantised scores), dropped candidates (-1), scores 0 and > 1, empty levels, a
    max_count that truncates, FCOS-sized and larger problems.  (A bucketed one-workgroup-per-image LDS variant passed
    this test but ran 3x slower end to end: FCOS scores crowd into a few score buckets, so it degenerates to O(n^2) on
    8 CUs instead of 256.)"""
    from oneshotdet_amd import ops
    rng = np.random.RandomState(7)
    if case == "levels_ties_drops":
        levels, topn, n_img, max_count = [(0, 700), (700, 300), (1000, 0), (1000, 41)], 250, 3, 500
    elif case == "single_segment":
        levels, topn, n_img, max_count = None, 0, 2, 2000
    elif case == "fcos_sized":
        levels, topn, n_img, max_count = [(0, 12800), (12800, 3200), (16000, 800), (16800, 208), (17008, 56)], 6000, 2, 10264
    else:
        levels, topn, n_img, max_count = [(0, 21000), (21000, 5000)], 9000, 1, 14000
    total = 2000 if levels is None else levels[-1][0] + levels[-1][1]
    scores = rng.rand(n_img, total).astype(np.float32) ** 3
    if case != "fcos_sized":
        scores = np.round(scores * 50) / 50                     # many exact ties
        scores[:, ::17] = -1.0                                  # dropped
        scores[:, 5::29] = 0.0
        scores[:, 3::31] = 1.5
    boxes = rng.rand(n_img, total, 4).astype(np.float32) * 100
    bs, ss, idx, cnt = ops.rank_sort_gather(torch.from_numpy(scores).cuda(), torch.from_numpy(boxes).cuda(), max_count,
                                            levels, topn)
    segs = levels if levels is not None else [(0, total)]
    cut = topn if levels is not None else total
    for i in range(n_img):
        keep = []
        for lo, c in segs:
            ids = np.arange(lo, lo + c)
            ids = ids[scores[i, ids] >= 0]
            order = np.lexsort((ids, -scores[i, ids]))          # score desc, then index asc
            keep.append(ids[order][:cut])
        keep = np.concatenate(keep) if keep else np.zeros((0,), np.int64)
        order = np.lexsort((keep, -scores[i, keep]))
        want = keep[order][:max_count]
        k = int(cnt[i])
        assert k == len(want), (case, i, k, len(want))
        np.testing.assert_array_equal(idx[i, :k].cpu().numpy(), want)
        np.testing.assert_array_equal(ss[i, :k].cpu().numpy(), scores[i, want])
        np.testing.assert_array_equal(bs[i, :k].cpu().numpy(), boxes[i, want])


@pytest.mark.parametrize("case", ["fcos_test", "fcos_train", "needs_full_order", "tiny", "ties"])
def test_proposals_sort_nms_equals_the_two_call_pipeline(case):
    """osd_proposals_sort_nms (ranks only the head of the score order, falls back to the full order per image when NMS
    cannot fill max_keep from it) against osd_rank_sort_gather + osd_nms_sorted: identical boxes, scores and counts.
    `needs_full_order`: boxes that nearly all overlap, so NMS exhausts the head and the second phase must run; `ties`:
    quantised scores (the threshold falls inside a run of equal keys); `tiny`: fewer candidates than the head."""
    from oneshotdet_amd import ops
    rng = np.random.RandomState(3)
    if case in ("fcos_test", "fcos_train", "needs_full_order", "ties"):
        levels = [(0, 12800), (12800, 3200), (16000, 800), (16800, 208), (17008, 56)]
        topn, max_keep = (12000, 4000) if case == "fcos_train" else (6000, 2000)
        n_img = 2
    else:
        levels, topn, max_keep, n_img = [(0, 300), (300, 100)], 250, 200, 3
    total = levels[-1][0] + levels[-1][1]
    scores = (rng.rand(n_img, total).astype(np.float32) ** 4) * 0.9 + 1e-4
    if case == "ties":
        scores = np.round(scores * 200) / 200
    scores[:, ::23] = -1.0
    ctr = rng.rand(n_img, total, 2).astype(np.float32) * np.array([1000.0, 780.0], np.float32)
    wh = rng.rand(n_img, total, 2).astype(np.float32) * 90 + 10
    if case == "needs_full_order":      # a few heavily overlapping clusters: NMS keeps very few boxes of the head
        ctr = (ctr // 250) * 250 + rng.rand(n_img, total, 2).astype(np.float32) * 6
        wh = np.full_like(wh, 200.0) + rng.rand(n_img, total, 2).astype(np.float32) * 4
        ctr[:, -3000:] = rng.rand(n_img, 3000, 2).astype(np.float32) * np.array([1000.0, 780.0], np.float32)   # low-score loners
        wh[:, -3000:] = 12.0
        scores[:, -3000:] = np.abs(scores[:, -3000:]) * 1e-3 + 1e-6
    boxes = np.concatenate([ctr - wh / 2, ctr + wh / 2], -1).astype(np.float32)
    max_count = sum(min(c, topn) for _, c in levels)
    k, b = torch.from_numpy(scores).cuda(), torch.from_numpy(boxes).cuda()
    for rule in (False, True):
        bs, ss, idx, cnt = ops.rank_sort_gather(k, b, max_count, levels, topn)
        rb, rs, _, rc = ops.nms_sorted(bs, ss, cnt, 0.8 if case != "needs_full_order" else 0.5, max_keep, cuda_semantics=rule)
        ob, os_, oc = ops.proposals_sort_nms(k, b, max_count, levels, topn, 0.8 if case != "needs_full_order" else 0.5,
                                             max_keep, cuda_semantics=rule)
        assert torch.equal(oc, rc), (case, oc.tolist(), rc.tolist())
        for i in range(n_img):
            c = int(rc[i])
            assert torch.equal(ob[i, :c], rb[i, :c]) and torch.equal(os_[i, :c], rs[i, :c]), (case, i)
        # any size of the exactly sorted head gives the same result; depth_out = how deep the scan read (position of the
        # last survivor + 1 in the score order), whichever phase produced the result
        _, _, rpos, _ = ops.nms_sorted(bs, ss, cnt, 0.8 if case != "needs_full_order" else 0.5, max_keep, cuda_semantics=rule)
        want_depth = [int(rpos[i, :int(rc[i])].max()) + 1 if int(rc[i]) else 0 for i in range(n_img)]
        for hint in (1, max_keep + max_keep // 2, max_count + 999):
            depth = torch.full((n_img,), -7, device="cuda", dtype=torch.int32)
            hb_, hs_, hc_ = ops.proposals_sort_nms(k, b, max_count, levels, topn, 0.8 if case != "needs_full_order" else 0.5,
                                                  max_keep, cuda_semantics=rule, head_hint=hint, depth_out=depth)
            assert torch.equal(hc_, rc) and depth.tolist() == want_depth, (case, hint, depth.tolist(), want_depth)
            for i in range(n_img):
                c = int(rc[i])
                assert torch.equal(hb_[i, :c], rb[i, :c]) and torch.equal(hs_[i, :c], rs[i, :c]), (case, hint, i)
    if case == "needs_full_order":      # the head (2564 candidates) did not fill max_keep: the second phase really ran
        hb, hs, _, hc = ops.nms_sorted(bs[:, :2564].contiguous(), ss[:, :2564].contiguous(), cnt.clamp(max=2564), 0.5, max_keep,
                                       cuda_semantics=True)
        assert int(hc.max()) < int(rc.min())


def test_osd_nms_single_entry_matches_the_oracle_for_any_scores():
    """osd_nms = _C.nms (csrc/nms.h:10) in ONE C-ABI call: ascending original indices + a device-side count; scores of
    either sign (the reference sorts whatever it is given), ties broken by the lower index."""
    from oneshotdet_amd import ops
    rng = np.random.RandomState(11)
    for n in (1, 63, 64, 65, 700, 5000):
        xy = rng.rand(n, 2).astype(np.float32) * 300
        wh = rng.rand(n, 2).astype(np.float32) * 120 + 1
        boxes = np.concatenate([xy, xy + wh], 1).astype(np.float32)
        scores = (rng.randn(n) * 3).astype(np.float32)              # negative scores included
        scores[rng.randint(0, n, max(1, n // 10))] = scores[0]      # ties
        for cuda in (True, False):
            keep, count = ops.nms(torch.from_numpy(boxes).cuda(), torch.from_numpy(scores).cuda(), 0.5, cuda_semantics=cuda)
            assert keep.dtype == torch.int64 and count.dtype == torch.int32 and keep.shape == (n,)
            got = keep[:int(count.item())].cpu().numpy()
            ref = orc.nms(boxes, scores, 0.5, cuda_semantics=cuda)      # stable argsort: ties -> lower index first
            assert np.array_equal(got, ref), (n, cuda)
    keep, count = ops.nms(torch.zeros(0, 4, device="cuda"), torch.zeros(0, device="cuda"), 0.5)
    assert int(count.item()) == 0 and keep.shape == (0,)


def test_proposal_depth_feedback_tracks_a_deepening_scan():
    """model.ProposalDepth: the lagged hint follows the scan depth of earlier calls (a trained head clusters its boxes, NMS
    reads deeper) without ever blocking, and never changes the result."""
    from oneshotdet_amd import model, ops
    rng = np.random.RandomState(5)
    levels, topn, max_keep = [(0, 12800), (12800, 3200), (16000, 1064)], 12000, 4000
    total = 17064
    scores = torch.from_numpy((rng.rand(2, total).astype(np.float32) ** 2) * 0.9 + 1e-4).cuda()
    ctr = (rng.rand(2, total, 2).astype(np.float32) * np.array([1000.0, 780.0], np.float32))
    depth = model.ProposalDepth()
    hints = []
    for spread in (90.0, 40.0, 20.0, 20.0, 20.0, 20.0):       # smaller boxes jitter -> more overlap -> deeper scans
        c2 = (ctr // spread) * spread
        boxes = torch.from_numpy(np.concatenate([c2 - 60, c2 + 60], -1).astype(np.float32)).cuda()
        d = depth.before(2, "cuda")
        out = ops.proposals_sort_nms(scores, boxes, total, levels, topn, 0.8, max_keep, cuda_semantics=True, head_hint=depth.hint,
                                     depth_out=d)
        depth.after(d)
        ref = ops.proposals_sort_nms(scores, boxes, total, levels, topn, 0.8, max_keep, cuda_semantics=True)
        assert all(torch.equal(a, b) for a, b in zip(out, ref))
        torch.cuda.synchronize()
        hints.append((depth.hint, d.tolist()))
    assert hints[-1][0] >= max(hints[-2][1]), hints          # the last call's head covered what the previous call needed
