"""Deterministic, platform-independent synthetic weights and inputs.

There is no network for checkpoints or datasets, so parity tests, golden fixtures and bench.py all use values
derived from an integer hash of (seed, tensor name, flat index): the same bits in the build container (where the
fixtures are generated from the real reference) and on the GPU box.  Only integer arithmetic plus one exact
uint32->float conversion and one multiply-add are involved, so no libm differences can creep in.

Distributions are chosen so activations stay O(1) through 50+ layers (the reference's
default init with identity FrozenBN blows features up to 1e2-1e3):
  conv weights       uniform(-b, b), b = sqrt(3 / fan_in)   (unit gain)
  FrozenBN weight    uniform(0.6, 1.4); bias, running_mean uniform(-0.3, 0.3); running_var uniform(0.6, 1.4)
  bn3 / downsample BN weight scaled by 0.5 so the residual stream does not grow
  GroupNorm weight   uniform(0.6, 1.4); bias uniform(-0.2, 0.2)
  head pred convs    b = 2 * sqrt(3 / fan_in); cls_logits bias = -log((1-p)/p) as in the reference (fcos.py:77-79)
  bbox_pred bias     uniform(2.0, 3.5) (boxes of tens of pixels, so NMS actually suppresses)
  scales             uniform(0.8, 1.2)
"""
import math
import zlib

import numpy as np


def _mix64(x):
    """splitmix64 finaliser on a uint64 array."""
    x = (x + np.uint64(0x9E3779B97F4A7C15))
    x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return x ^ (x >> np.uint64(31))


def uniform01(name, n, seed=0):
    """n float32 values in [0, 1) determined by (seed, name, index)."""
    with np.errstate(over="ignore"):
        base = np.uint64(zlib.crc32(name.encode()) & 0xFFFFFFFF) * np.uint64(0x100000001B3) + np.uint64(seed)
        idx = np.arange(n, dtype=np.uint64)
        h = _mix64(_mix64(idx + base * np.uint64(0x2545F4914F6CDD1D)) ^ base)
    top24 = (h >> np.uint64(40)).astype(np.uint32)
    return top24.astype(np.float32) * np.float32(1.0 / (1 << 24))


def uniform(name, shape, lo, hi, seed=0):
    n = int(np.prod(shape)) if len(shape) else 1
    u = uniform01(name, n, seed)
    return (u * np.float32(hi - lo) + np.float32(lo)).astype(np.float32).reshape(shape)


def _box_head_kind(key, shape):
    """roi_heads.box.* (spec.box_head_shapes): Sequential index 1 / 4 = GroupNorm, 2-D weights = Linear."""
    leaf = key.rsplit(".", 1)[-1]
    if "predictor.cls_score" in key:
        return "box_cls_w" if leaf == "weight" else "box_cls_b"
    if "predictor.bbox_pred" in key:
        return "box_reg_w" if leaf == "weight" else "box_reg_b"
    if len(shape) in (2, 4):
        return "conv_w"
    parts = key.split(".")
    if parts[-2].isdigit() and int(parts[-2]) in (1, 4):
        return "gn_" + leaf
    return "conv_b"


def _kind(key, shape):
    leaf = key.rsplit(".", 1)[-1]
    if key.startswith("roi_heads."):
        return _box_head_kind(key, shape)
    if key.endswith(".scale"):
        return "scale"
    if len(shape) == 4:
        if any(s in key for s in ("cls_logits", "bbox_pred", "centerness")):
            return "pred_w"
        return "conv_w"
    if ".bn" in key or "downsample.1." in key:
        return "bn_" + leaf
    if "tower" in key:
        # Sequential index % 3: 0 = conv bias, 1 = GroupNorm
        idx = int(key.split(".")[-2])
        return ("gn_" + leaf) if idx % 3 == 1 else "conv_b"
    if key.endswith("cls_logits.bias"):
        return "cls_bias"
    if key.endswith("bbox_pred.bias"):
        return "reg_bias"
    return "conv_b"


def make_tensor(key, shape, seed=0):
    kind = _kind(key, shape)
    if kind == "box_cls_w":     # class scores spread over a few units so NMS sees a real ranking
        return uniform(key, shape, -0.08, 0.08, seed)
    if kind == "box_reg_w":     # deltas / (10, 10, 5, 5): shifts of a few percent of the proposal size
        return uniform(key, shape, -0.04, 0.04, seed)
    if kind in ("box_cls_b", "box_reg_b"):
        return uniform(key, shape, -0.1, 0.1, seed)
    if kind in ("conv_w", "pred_w"):
        fan_in = int(np.prod(shape[1:]))
        b = math.sqrt(3.0 / fan_in) * (2.0 if kind == "pred_w" else 1.0)
        return uniform(key, shape, -b, b, seed)
    if kind == "bn_weight":
        w = uniform(key, shape, 0.6, 1.4, seed)
        if ".bn3." in key or "downsample.1." in key:
            w = w * np.float32(0.5)
        return w
    if kind in ("bn_bias", "bn_running_mean"):
        return uniform(key, shape, -0.3, 0.3, seed)
    if kind == "bn_running_var":
        return uniform(key, shape, 0.6, 1.4, seed)
    if kind == "gn_weight":
        return uniform(key, shape, 0.6, 1.4, seed)
    if kind == "gn_bias":
        return uniform(key, shape, -0.2, 0.2, seed)
    if kind == "conv_b":
        return uniform(key, shape, -0.1, 0.1, seed)
    if kind == "cls_bias":
        from .spec import PRIOR_PROB
        return np.full(shape, -math.log((1 - PRIOR_PROB) / PRIOR_PROB), dtype=np.float32)
    if kind == "reg_bias":      # exp(~3) = 20 px distances so neighbouring boxes overlap and NMS has work to do
        return uniform(key, shape, 2.0, 3.5, seed)
    if kind == "scale":
        return uniform(key, shape, 0.8, 1.2, seed)
    raise KeyError(key)


def make_state_dict(shapes, seed=0):
    """{name: np.float32 array} for an OrderedDict of {name: shape} (see spec.hot_path_shapes)."""
    return {k: make_tensor(k, tuple(s), seed) for k, s in shapes.items()}


def make_images(name, batch, h, w, seed=0, scale=50.0):
    """Synthetic BGR255-minus-mean-like images: uniform with the std of N(0,1)*50."""
    half = scale * math.sqrt(3.0)
    return uniform("input." + name, (batch, 3, h, w), -half, half, seed)


def make_gt_boxes(batch, h, w, seed=0, max_boxes=6):
    """Per image 1..max_boxes boxes with side in [32, 512] px clipped to the image, label 1 (SURVEY.md §8d
    config 3).  Returns a list of float32 [n, 4] xyxy arrays."""
    out = []
    for i in range(batch):
        u = uniform01("gt.%d" % i, 1 + 4 * max_boxes, seed)
        n = 1 + int(u[0] * max_boxes) % max_boxes
        boxes = np.zeros((n, 4), dtype=np.float32)
        for j in range(n):
            bw = 32 + u[1 + 4 * j] * min(480, w - 33)
            bh = 32 + u[2 + 4 * j] * min(480, h - 33)
            x1 = u[3 + 4 * j] * (w - 1 - bw)
            y1 = u[4 + 4 * j] * (h - 1 - bh)
            boxes[j] = (math.floor(x1), math.floor(y1), math.floor(x1 + bw), math.floor(y1 + bh))
        out.append(boxes)
    return out
