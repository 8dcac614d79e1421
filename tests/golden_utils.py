"""Helpers shared by tests/golden/make_golden.py (fixture writer) and the parity tests (fixture readers)."""
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
N_SAMPLES = 4096


def sample_indices(numel, tag):
    """Deterministic pseudo-random flat indices (with replacement) into a tensor of `numel` elements."""
    from oneshotdet_amd.synth import uniform01
    u = uniform01("sample." + tag, N_SAMPLES, seed=7).astype(np.float64)
    return np.minimum((u * numel).astype(np.int64), numel - 1)


def checksum(arr, tag):
    """Compact summary of a [B, C, H, W] float tensor: per-channel mean / absmax + hashed sample points."""
    a = np.asarray(arr, dtype=np.float32)
    flat = a.reshape(-1)
    return {
        tag + ".shape": np.asarray(a.shape, dtype=np.int64),
        tag + ".mean": a.astype(np.float64).mean(axis=(0, 2, 3)).astype(np.float32),
        tag + ".absmax": np.abs(a).max(axis=(0, 2, 3)),
        tag + ".samples": flat[sample_indices(flat.size, tag)],
    }


def check_against(arr, fixture, tag, atol_rel=1e-3, rtol=1e-3):
    """Assert `arr` matches the checksum stored under `tag` (features: atol = atol_rel * absmax, rtol)."""
    a = np.asarray(arr, dtype=np.float32)
    assert tuple(fixture[tag + ".shape"]) == a.shape, (tag, a.shape, fixture[tag + ".shape"])
    absmax = float(fixture[tag + ".absmax"].max())
    atol = atol_rel * absmax
    flat = a.reshape(-1)
    got = flat[sample_indices(flat.size, tag)]
    np.testing.assert_allclose(got, fixture[tag + ".samples"], rtol=rtol, atol=atol, err_msg=tag + " samples")
    np.testing.assert_allclose(a.astype(np.float64).mean(axis=(0, 2, 3)), fixture[tag + ".mean"], rtol=rtol,
                               atol=atol, err_msg=tag + " mean")
    np.testing.assert_allclose(np.abs(a).max(axis=(0, 2, 3)), fixture[tag + ".absmax"], rtol=rtol, atol=atol,
                               err_msg=tag + " absmax")


def flatten_head(logits, bbox_reg, centerness):
    """list of [B,C,H,W] per level -> [B, sum(HW), 6] (logit, l, t, r, b, ctr) in level-major, row-major order."""
    outs = []
    for lg, br, ct in zip(logits, bbox_reg, centerness):
        lg, br, ct = (np.asarray(t, dtype=np.float32) for t in (lg, br, ct))
        B = lg.shape[0]
        outs.append(np.concatenate([lg.reshape(B, 1, -1), br.reshape(B, 4, -1), ct.reshape(B, 1, -1)], axis=1)
                    .transpose(0, 2, 1))
    return np.concatenate(outs, axis=1)


def load(name):
    return np.load(os.path.join(GOLDEN_DIR, name))


def match_boxes(boxes_a, scores_a, boxes_b, scores_b, tol=1e-2):
    """Fraction of (box, score) rows of A that have a counterpart in B (order-free; scores sorted then matched)."""
    if len(boxes_a) == 0:
        return 1.0 if len(boxes_b) == 0 else 0.0
    ra = np.concatenate([boxes_a, scores_a[:, None] * 1000.0], axis=1)
    rb = np.concatenate([boxes_b, scores_b[:, None] * 1000.0], axis=1)
    hits = 0
    from scipy.spatial import cKDTree
    tree = cKDTree(rb)
    d, _ = tree.query(ra, k=1)
    hits = int((d <= tol * 5).sum())
    return hits / float(len(ra))


# name -> (batch, H, W, shots, query_h, query_w)
CASES = {
    "small": (1, 128, 160, 1, 63, 63),          # full tensors
    "config1": (1, 800, 1024, 1, 127, 127),     # BASELINE.json configs[0]
    "nonsquare": (2, 96, 160, 1, 96, 160),      # pins the (h, w)-as-(x2, y2) query-box quirk
    "shots5": (2, 128, 128, 5, 127, 127),       # S=5 mean-pooled queries (configs[4])
    "tall": (1, 160, 96, 1, 64, 96),
    # two DISTINCT full-size (target, query, boxes) triples: the bs=8 benchmark batch built from them is checked per image
    # against this (cross-image aliasing at the benchmark's grid sizes)
    "config1x2": (2, 800, 1024, 1, 127, 127),
    # BASELINE.json configs[4] / SURVEY.md 8c case (v): the multi-scale shapes (short edge 640 and 1024, long edge = short x
    # 1.28 rounded up to /32) with the 5-shot query stack; short edge 800 with one shot is `config1`
    "ms640": (1, 640, 832, 5, 127, 127),
    "ms1024": (1, 1024, 1312, 5, 127, 127),
}


def case_inputs(name, seed=0):
    """(images [B,3,H,W], queries [B*S,3,h,w]) float32 numpy for a named case."""
    from oneshotdet_amd import synth
    B, H, W, S, qh, qw = CASES[name]
    return (synth.make_images("target." + name, B, H, W, seed), synth.make_images("query." + name, B * S, qh, qw, seed))


# R0 (to_image_list on LISTS of different-size images, structures/image_list.py:52-70): target sizes, query sizes
RAGGED = {"targets": [(96, 160), (128, 130)], "queries": [(63, 63), (96, 80)], "size_divisible": 32}


def ragged_inputs(seed=0):
    """lists of CHW float32 numpy arrays: (targets, queries) of the `ragged` case."""
    from oneshotdet_amd import synth
    t = [synth.make_images("target.ragged.%d" % i, 1, h, w, seed)[0] for i, (h, w) in enumerate(RAGGED["targets"])]
    q = [synth.make_images("query.ragged.%d" % i, 1, h, w, seed)[0] for i, (h, w) in enumerate(RAGGED["queries"])]
    return t, q


def voc_eval_inputs(seed=0, n_images=7, max_det=48, max_gt=9, n_classes=3):
    """Synthetic detections / ground truth for the VOC evaluation fixtures (tests/golden/voc_eval.npz): per image
    (pred boxes [n,4], labels [n], scores [n]) and (gt boxes [g,4], labels [g], difficult [g]); detections are jittered copies
    of ground-truth boxes (several per box: duplicates), plus false positives; distinct scores (ties are order-dependent in
    the reference's argsort); an image without ground truth of one class, one without detections of a class, difficult boxes."""
    from oneshotdet_amd.synth import uniform01
    preds, gts = [], []
    for i in range(n_images):
        u = uniform01("voc.%d" % i, 4096, seed).astype(np.float64)
        k = 0

        def nxt(n):
            nonlocal k
            v = u[k:k + n]
            k += n
            return v
        g = 1 + int(nxt(1)[0] * (max_gt - 1)) if i != 3 else 2
        xy = nxt(2 * g).reshape(g, 2) * 400.0
        wh = 20.0 + nxt(2 * g).reshape(g, 2) * 180.0
        gb = np.concatenate([xy, xy + wh], 1).round().astype(np.float32)
        gl = (1 + (nxt(g) * n_classes).astype(np.int64)).clip(1, n_classes)
        gd = (nxt(g) < 0.2).astype(np.uint8)
        if i == 3:
            gl[:] = 1                                   # no ground truth of classes 2, 3 in this image
        n = int(nxt(1)[0] * max_det) if i != 5 else 0    # image 5: no detections at all
        src = (nxt(n) * g).astype(np.int64).clip(0, g - 1)
        jit = (nxt(4 * n).reshape(n, 4) - 0.5) * 60.0
        pb = gb[src] + jit.astype(np.float32)
        far = nxt(n) < 0.25
        pb[far] += 300.0
        pb[:, 2:] = np.maximum(pb[:, 2:], pb[:, :2] + 1.0)
        pl = gl[src].copy()
        swap = nxt(n) < 0.15
        pl[swap] = 1 + (pl[swap] % n_classes)
        if i == 1:
            pl[pl == 2] = 3                              # image 1: no detections of class 2
        ps = (nxt(n) * 0.98 + 0.01 + np.arange(n) * 1e-4).astype(np.float32)
        preds.append((pb.astype(np.float32), pl.astype(np.int64), ps))
        gts.append((gb, gl.astype(np.int64), gd))
    return preds, gts


def dataset_inputs():
    """A small synthetic COCO-format annotation set + images for the few-shot dataset fixture (tests/golden/dataset.npz):
    14 images of 40..120 pixels a side, 4 categories with non-contiguous json ids (1, 3, 7, 9), 2 - 5 annotations per image with
    crowd flags, boxes that reach past the image, degenerate (width <= 1) boxes, objects below the support-area threshold, and
    one image whose only object of a category is a crowd.  Returns (coco dict, {image id: uint8 HxWx3})."""
    rs = np.random.RandomState(20241003)
    cats = [{"id": 1, "name": "a"}, {"id": 3, "name": "b"}, {"id": 7, "name": "c"}, {"id": 9, "name": "d"}]
    images, anns, pix = [], [], {}
    aid = 100
    for k in range(14):
        iid = 1000 + 7 * k
        h, w = int(rs.randint(40, 121)), int(rs.randint(40, 121))
        images.append({"id": iid, "file_name": "img_%d.png" % iid, "height": h, "width": w})
        pix[iid] = rs.randint(0, 256, size=(h, w, 3)).astype(np.uint8)
        for j in range(int(rs.randint(2, 6))):
            cat = [1, 3, 7, 9][int(rs.randint(0, 4))]
            bw, bh = float(rs.uniform(1.0, w * 0.9)), float(rs.uniform(1.0, h * 0.9))
            if rs.rand() < 0.12:
                bw = float(rs.uniform(0.2, 1.0))                    # "close to zero area" (has_valid_annotation: any side <= 1)
            x, y = float(rs.uniform(-3.0, w - 2.0)), float(rs.uniform(-3.0, h - 2.0))      # may start outside / reach past the image
            anns.append({"id": aid, "image_id": iid, "category_id": cat, "bbox": [round(x, 2), round(y, 2), round(bw, 2), round(bh, 2)],
                         "area": round(bw * bh, 2), "iscrowd": int(rs.rand() < 0.15)})
            aid += 1
    # an image whose only category-9 object is a crowd: it must not enter category 9's catalog
    images.append({"id": 5000, "file_name": "img_5000.png", "height": 64, "width": 80})
    pix[5000] = rs.randint(0, 256, size=(64, 80, 3)).astype(np.uint8)
    anns.append({"id": aid, "image_id": 5000, "category_id": 9, "bbox": [5.0, 6.0, 40.0, 30.0], "area": 1200.0, "iscrowd": 1})
    anns.append({"id": aid + 1, "image_id": 5000, "category_id": 1, "bbox": [10.5, 3.5, 33.3, 44.4], "area": 1478.52, "iscrowd": 0})
    return {"images": images, "annotations": anns, "categories": cats}, pix


# dataset fixture configurations: (name, is_train, shots, support augmentation (0 or 1 = + horizontal flip), training-excluded
# contiguous category ids, test-excluded ones, support-area threshold)
DATASET_CONFIGS = [("train1", True, 1, 0, [2], [], 150.0), ("train3aug", True, 3, 1, [], [], 60.0), ("test1", False, 1, 0, [2], [3], 150.0)]
