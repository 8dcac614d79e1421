"""ORACLE — test infrastructure only.  CPU restatement of the reference's siamese-FCOS hot path.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this module; the
product (`oneshotdet_amd/`) never does and fails loudly when its HIP library is missing.

Every function restates one reference function in plain PyTorch-CPU fp32 (NCHW, like the reference) and cites the
file:line it follows (paths relative to /root/reference/maskrcnn_benchmark/).  The restatement is pinned against
the REAL reference, imported in the build container by tests/golden/ref_harness.py: tests/golden/make_golden.py
runs both on the same synthetic weights/inputs, asserts agreement, and writes the golden fixtures that
tests/test_oracle_golden.py re-checks everywhere (including the GPU box, where /root/reference is absent).
NMS is additionally pinned by the reference's own known-answer vectors (tests/test_nms.py, via
tests/golden/nms_kat.npz).

Arithmetic that lives outside /root/reference: conv2d, group_norm, max_pool2d, interpolate, sigmoid, exp, topk, sort
are ATen (PyTorch); the reference pins none of them with tests, so they are pinned only by the goldens.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

FPN_STRIDES = (8, 16, 32, 64, 128)
POOLER_SCALES = (0.125, 0.0625, 0.03125, 0.015625, 0.0078125)
INF = 100000000


# --------------------------------------------------------------------------------------------------------------------
# reduced-precision emulation (what the bf16 engines store) — `emu=None` everywhere below is the plain fp32 restatement
# --------------------------------------------------------------------------------------------------------------------
class _RoundBothWays(torch.autograd.Function):
    """A tensor the engine STORES in `dtype`: the value is rounded on the way forward and its gradient — which the engine
    also stores in `dtype` — on the way back (the sum over all consumers is formed in fp32 and rounded once)."""

    @staticmethod
    def forward(ctx, x, dtype):
        ctx.dtype = dtype
        return x.to(dtype).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        return g.to(ctx.dtype).to(g.dtype), None


class Emulation(object):
    """Reduced-precision mode of this oracle: the SAME functions, with every tensor the bf16 engines write to HBM rounded
    where they round it — the packed image, every conv output after its fused epilogue (bias / FrozenBN shift, residual,
    ReLU or exp), correlation, GroupNorm+ReLU outputs, the head outputs, and on the way back every stored gradient — and
    with FrozenBN folded into the conv rows BEFORE the weights are rounded (w' = round(w * scale), shift kept in fp32:
    `osd_pack_conv_weight`).  Accumulation stays fp32, as in the MFMA kernels.  dtype=None switches the rounding off and
    keeps only the folded form: that must reproduce the plain restatement to fp32 rounding (tests/test_oracle_golden.py).
    fused_downsample: stage names whose first block computes conv3 + downsample as ONE GEMM (a single rounding of the sum):
    the training engine fuses the frozen layer1 only, the inference engine all four stages."""

    def __init__(self, dtype=torch.bfloat16, fused_downsample=("layer1",)):
        self.dtype = dtype
        self.fused_downsample = tuple(fused_downsample)

    def act(self, x):
        return x if self.dtype is None else _RoundBothWays.apply(x, self.dtype)

    def weight(self, w):
        """Packed weights are rounded copies of the fp32 masters; the master's gradient is not rounded (fp32 dW)."""
        if self.dtype is None:
            return w
        return w + (w.to(self.dtype).to(w.dtype) - w).detach()


def _conv_bn(x, sd, conv, bn, emu, stride=1, padding=0):
    """conv -> FrozenBN.  emu: the folded form the kernels compute (scale into the weight rows, shift as the bias)."""
    if emu is None:
        return frozen_bn(F.conv2d(x, sd[conv + ".weight"], None, stride=stride, padding=padding), sd, bn)
    scale = sd[bn + ".weight"] * sd[bn + ".running_var"].rsqrt()
    shift = sd[bn + ".bias"] - sd[bn + ".running_mean"] * scale
    return F.conv2d(x, emu.weight(sd[conv + ".weight"] * scale.reshape(-1, 1, 1, 1)), shift, stride=stride, padding=padding)


# --------------------------------------------------------------------------------------------------------------------
# backbone
# --------------------------------------------------------------------------------------------------------------------
def frozen_bn(x, sd, p):
    """layers/batch_norm.py:19-24 — note: NO epsilon."""
    scale = sd[p + ".weight"] * sd[p + ".running_var"].rsqrt()
    bias = sd[p + ".bias"] - sd[p + ".running_mean"] * scale
    return x * scale.reshape(1, -1, 1, 1) + bias.reshape(1, -1, 1, 1)


def stem(x, sd, p, emu=None):
    """modeling/backbone/resnet.py:332-337 (BaseStem.forward)."""
    if emu is not None:
        x = emu.act(F.relu(_conv_bn(emu.act(x), sd, p + "conv1", p + "bn1", emu, stride=2, padding=3)))
        return F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
    x = F.conv2d(x, sd[p + "conv1.weight"], None, stride=2, padding=3)
    x = F.relu(frozen_bn(x, sd, p + "bn1"))
    return F.max_pool2d(x, kernel_size=3, stride=2, padding=1)


def bottleneck(x, sd, p, stride, emu=None):
    """modeling/backbone/resnet.py:295-315 (Bottleneck.forward); stride sits in the 1x1 conv1 and in the
    downsample 1x1 (STRIDE_IN_1X1 True, resnet.py:245-263)."""
    if emu is not None:
        o1 = emu.act(F.relu(_conv_bn(x, sd, p + "conv1", p + "bn1", emu, stride=stride)))
        o2 = emu.act(F.relu(_conv_bn(o1, sd, p + "conv2", p + "bn2", emu, padding=1)))
        out = _conv_bn(o2, sd, p + "conv3", p + "bn3", emu)                 # fp32 accumulators: not stored
        identity = x
        if (p + "downsample.0.weight") in sd:
            identity = _conv_bn(x, sd, p + "downsample.0", p + "downsample.1", emu, stride=stride)
            if not any(("." + st + ".") in ("." + p) for st in emu.fused_downsample):
                identity = emu.act(identity)                                # its own launch: stored, then read back
        return emu.act(F.relu(out + identity))
    identity = x
    out = F.conv2d(x, sd[p + "conv1.weight"], None, stride=stride)
    out = F.relu(frozen_bn(out, sd, p + "bn1"))
    out = F.conv2d(out, sd[p + "conv2.weight"], None, stride=1, padding=1)
    out = F.relu(frozen_bn(out, sd, p + "bn2"))
    out = frozen_bn(F.conv2d(out, sd[p + "conv3.weight"], None), sd, p + "bn3")
    if (p + "downsample.0.weight") in sd:
        identity = frozen_bn(F.conv2d(x, sd[p + "downsample.0.weight"], None, stride=stride), sd,
                             p + "downsample.1")
    return F.relu(out + identity)


def resnet_body(x, sd, p, blocks=(3, 4, 6, 3), emu=None):
    """modeling/backbone/resnet.py:138-145 (ResNet.forward) -> [C2, C3, C4, C5]."""
    x = stem(x, sd, p + "stem.", emu)
    outs = []
    for si, n in enumerate(blocks):
        for b in range(n):
            x = bottleneck(x, sd, "%slayer%d.%d." % (p, si + 1, b), stride=2 if (b == 0 and si > 0) else 1, emu=emu)
        outs.append(x)
    return outs


def fpn(feats, sd, p, emu=None):
    """modeling/backbone/fpn.py:43-75 + LastLevelP6P7.forward :95-99.  C2 is ignored (fpn.py:33); P6 comes from
    P5 because in_channels == out_channels (USE_C5 False; fpn.py:93-96); P7 sees relu(P6)."""
    c3, c4, c5 = feats[1], feats[2], feats[3]

    def conv(name, x, **kw):
        return F.conv2d(x, sd[p + name + ".weight"], sd[p + name + ".bias"], **kw)

    if emu is not None:       # every FPN conv is one launch: output (+ the top-down addend, fused) rounded once
        def econv(name, x, add=None, **kw):
            y = F.conv2d(x, emu.weight(sd[p + name + ".weight"]), sd[p + name + ".bias"], **kw)
            return emu.act(y if add is None else y + add)
        inner4 = econv("fpn_inner4", c5)
        p5 = econv("fpn_layer4", inner4, padding=1)
        inner3 = econv("fpn_inner3", c4, add=F.interpolate(inner4, scale_factor=2, mode="nearest"))
        p4 = econv("fpn_layer3", inner3, padding=1)
        inner2 = econv("fpn_inner2", c3, add=F.interpolate(inner3, scale_factor=2, mode="nearest"))
        p3 = econv("fpn_layer2", inner2, padding=1)
        p6 = econv("top_blocks.p6", p5, stride=2, padding=1)
        p7 = econv("top_blocks.p7", F.relu(p6), stride=2, padding=1)
        return [p3, p4, p5, p6, p7]
    inner4 = conv("fpn_inner4", c5)
    p5 = conv("fpn_layer4", inner4, padding=1)
    inner3 = conv("fpn_inner3", c4) + F.interpolate(inner4, scale_factor=2, mode="nearest")
    p4 = conv("fpn_layer3", inner3, padding=1)
    inner2 = conv("fpn_inner2", c3) + F.interpolate(inner3, scale_factor=2, mode="nearest")
    p3 = conv("fpn_layer2", inner2, padding=1)
    p6 = conv("top_blocks.p6", p5, stride=2, padding=1)
    p7 = conv("top_blocks.p7", F.relu(p6), stride=2, padding=1)
    return [p3, p4, p5, p6, p7]


def backbone(x, sd, p, emu=None):
    """modeling/backbone/backbone.py:51-72: Sequential(body, fpn)."""
    return fpn(resnet_body(x, sd, p + "body.", emu=emu), sd, p + "fpn.", emu)


# --------------------------------------------------------------------------------------------------------------------
# query pooling + correlation
# --------------------------------------------------------------------------------------------------------------------
def _bilinear_taps(y, x, height, width):
    """csrc/cpu/ROIAlign_cpu.cpp:44-104 (pre_calc_for_bilinear_interpolate), one sample point."""
    if y < -1.0 or y > height or x < -1.0 or x > width:
        return None
    y = max(y, 0.0)
    x = max(x, 0.0)
    y_low, x_low = int(y), int(x)
    if y_low >= height - 1:
        y_high = y_low = height - 1
        y = float(y_low)
    else:
        y_high = y_low + 1
    if x_low >= width - 1:
        x_high = x_low = width - 1
        x = float(x_low)
    else:
        x_high = x_low + 1
    ly, lx = np.float32(y) - np.float32(y_low), np.float32(x) - np.float32(x_low)
    hy, hx = np.float32(1.0) - ly, np.float32(1.0) - lx
    return ((y_low, x_low, hy * hx), (y_low, x_high, hy * lx), (y_high, x_low, ly * hx), (y_high, x_high, ly * lx))


def roi_align(inp, rois, spatial_scale, ph, pw, sampling_ratio):
    """csrc/cpu/ROIAlign_cpu.cpp:114-219 (ROIAlignForward_cpu_kernel), float32, differentiable w.r.t. `inp`
    (the tap weights are constants; gradient math = csrc/cuda/ROIAlign_cuda.cu:178-254).
    inp [B,C,H,W]; rois [R,5] = (batch_idx, x1, y1, x2, y2)."""
    B, C, H, W = inp.shape
    out = []
    f32 = np.float32
    for r in rois.detach().cpu().numpy().astype(np.float32):
        bi = int(r[0])
        scale = f32(spatial_scale)
        rsw, rsh, rew, reh = r[1] * scale, r[2] * scale, r[3] * scale, r[4] * scale
        roi_w = max(rew - rsw, f32(1.0))
        roi_h = max(reh - rsh, f32(1.0))
        bin_h, bin_w = f32(roi_h) / f32(ph), f32(roi_w) / f32(pw)
        gh = sampling_ratio if sampling_ratio > 0 else int(math.ceil(roi_h / ph))
        gw = sampling_ratio if sampling_ratio > 0 else int(math.ceil(roi_w / pw))
        count = f32(gh * gw)
        cells = []
        for i in range(ph):
            for j in range(pw):
                acc = inp.new_zeros((C,))
                for iy in range(gh):
                    yy = rsh + f32(i) * bin_h + f32(iy + 0.5) * bin_h / f32(gh)
                    for ix in range(gw):
                        xx = rsw + f32(j) * bin_w + f32(ix + 0.5) * bin_w / f32(gw)
                        taps = _bilinear_taps(float(yy), float(xx), H, W)
                        if taps is None:
                            continue
                        # ROIAlign_cpu.cpp:199-202: one expression w1*d1 + w2*d2 + w3*d3 + w4*d4 added to the sum
                        t = None
                        for (ty, tx, wgt) in taps:
                            term = float(wgt) * inp[bi, :, ty, tx]
                            t = term if t is None else t + term
                        acc = acc + t
                cells.append(acc / float(count))
        out.append(torch.stack(cells, dim=1).reshape(C, ph, pw))
    return torch.stack(out, dim=0)


def roi_align_backward_cuda(grad, rois, spatial_scale, ph, pw, batch, channels, height, width, sampling_ratio):
    """The reference's ROIAlign BACKWARD, which exists on CUDA only (csrc/ROIAlign.h:27-46 raises on the CPU): a restatement of
    RoIAlignBackwardFeature, csrc/cuda/ROIAlign_cuda.cu:178-254, with its bilinear_interpolate_gradient (:125-176), in float32 —
    per pooled cell and sample point g_k = top_diff * w_k / count scattered into the four taps.  It cannot be run against the
    kernel here (no CUDA); tests/test_oracle_golden.py checks that autograd through `roi_align` above — what the query-branch
    gradient fixtures were recorded with — equals this statement of the kernel's arithmetic.
    grad [R, C, ph, pw] float32 numpy; rois [R, 5]; -> bottom_diff [batch, C, H, W] float32."""
    f32 = np.float32
    out = np.zeros((batch, channels, height, width), np.float32)
    for n, r in enumerate(np.asarray(rois, np.float32)):
        bi = int(r[0])
        scale = f32(spatial_scale)
        rsw, rsh, rew, reh = r[1] * scale, r[2] * scale, r[3] * scale, r[4] * scale          # :196-199 (no rounding)
        roi_w, roi_h = max(rew - rsw, f32(1.0)), max(reh - rsh, f32(1.0))                      # :206-207
        bin_h, bin_w = f32(roi_h) / f32(ph), f32(roi_w) / f32(pw)
        gh = sampling_ratio if sampling_ratio > 0 else int(math.ceil(roi_h / ph))              # :218-219
        gw = sampling_ratio if sampling_ratio > 0 else int(math.ceil(roi_w / pw))
        count = f32(gh * gw)
        for i in range(ph):
            for j in range(pw):
                top = grad[n, :, i, j].astype(np.float32)
                for iy in range(gh):
                    y = rsh + f32(i) * bin_h + f32(iy + 0.5) * bin_h / f32(gh)                 # :226
                    for ix in range(gw):
                        x = rsw + f32(j) * bin_w + f32(ix + 0.5) * bin_w / f32(gw)             # :229
                        y_, x_ = f32(y), f32(x)
                        if y_ < -1.0 or y_ > height or x_ < -1.0 or x_ > width:                # :134-139: no contribution
                            continue
                        y_, x_ = max(y_, f32(0.0)), max(x_, f32(0.0))
                        y_low, x_low = int(y_), int(x_)
                        if y_low >= height - 1:
                            y_high = y_low = height - 1
                            y_ = f32(y_low)
                        else:
                            y_high = y_low + 1
                        if x_low >= width - 1:
                            x_high = x_low = width - 1
                            x_ = f32(x_low)
                        else:
                            x_high = x_low + 1
                        ly, lx = f32(y_) - f32(y_low), f32(x_) - f32(x_low)
                        hy, hx = f32(1.0) - ly, f32(1.0) - lx
                        for (ty, tx, w) in ((y_low, x_low, hy * hx), (y_low, x_high, hy * lx), (y_high, x_low, ly * hx), (y_high, x_high, ly * lx)):
                            out[bi, :, ty, tx] += top * f32(w) / count                          # :238-248 (atomicAdd of g_k)
    return out


def query_boxes(image_sizes):
    """modeling/detector/generalized_rcnn.py:257 + SuppAlignLayer.convert_to_roi_format :33-45.
    Quirk: the box is [0, 0, h, w] built from image_sizes = (h, w) but consumed as (x1, y1, x2, y2)."""
    rois = [[float(i), 0.0, 0.0, float(h), float(w)] for i, (h, w) in enumerate(image_sizes)]
    return torch.tensor(rois, dtype=torch.float32)


def query_pool(query_feats, image_sizes, batch_size):
    """SuppAlignLayer.forward generalized_rcnn.py:47-52 (ROIAlign (1,1), scale per level, sampling_ratio 2)
    followed by batch_pooling :100-104 (mean over the shots of each target image). -> 5 x [B, C, 1, 1]."""
    rois = query_boxes(image_sizes)
    pooled = []
    for feat, scale in zip(query_feats, POOLER_SCALES):
        v = roi_align(feat, rois, scale, 1, 1, 2)
        D, C, Hh, Ww = v.shape
        pooled.append(v.view(batch_size, D // batch_size, C, Hh, Ww).mean(dim=1))
    return pooled


def correlate(target_feats, pooled, emu=None):
    """generalized_rcnn.py:307-311: features[i] * pooled[i].expand(-1, -1, H, W) — depthwise x-corr, 1x1 kernel."""
    out = [f * q.expand(-1, -1, f.shape[2], f.shape[3]) for f, q in zip(target_feats, pooled)]
    return out if emu is None else [emu.act(t) for t in out]


# --------------------------------------------------------------------------------------------------------------------
# FCOS head
# --------------------------------------------------------------------------------------------------------------------
def fcos_head(feats, sd, p="rpn.head.", emu=None):
    """modeling/rpn/fcos/fcos.py:83-99 (FCOSHead.forward).  Shared weights across levels; centerness from the CLS
    tower (:92); bbox_reg = exp(scale_l * bbox_pred(bbox_tower)) (:95-97).  emu: the conv output is stored (rounded) before
    GroupNorm reads it, GroupNorm+ReLU is one more stored tensor, the prediction convs store their (exp'd) outputs."""
    rw = (lambda w: w) if emu is None else emu.weight
    ra = (lambda t: t) if emu is None else emu.act

    def tower(x, name):
        for i in range(4):
            x = ra(F.conv2d(x, rw(sd["%s%s.%d.weight" % (p, name, 3 * i)]), sd["%s%s.%d.bias" % (p, name, 3 * i)],
                            padding=1))
            x = F.group_norm(x, 32, sd["%s%s.%d.weight" % (p, name, 3 * i + 1)],
                             sd["%s%s.%d.bias" % (p, name, 3 * i + 1)], eps=1e-5)
            x = ra(F.relu(x))
        return x

    logits, bbox_reg, centerness = [], [], []
    for l, f in enumerate(feats):
        ct = tower(f, "cls_tower")
        logits.append(ra(F.conv2d(ct, rw(sd[p + "cls_logits.weight"]), sd[p + "cls_logits.bias"], padding=1)))
        centerness.append(ra(F.conv2d(ct, rw(sd[p + "centerness.weight"]), sd[p + "centerness.bias"], padding=1)))
        bt = tower(f, "bbox_tower")
        bp = F.conv2d(bt, rw(sd[p + "bbox_pred.weight"]), sd[p + "bbox_pred.bias"], padding=1)
        bbox_reg.append(ra(torch.exp(bp * sd["%sscales.%d.scale" % (p, l)])))
    return logits, bbox_reg, centerness


def compute_locations(level_hw):
    """fcos.py:209-234: (x, y) = (j*s + s//2, i*s + s//2), row-major, float32."""
    locs = []
    for (h, w), s in zip(level_hw, FPN_STRIDES):
        xs = torch.arange(0, w * s, step=s, dtype=torch.float32)
        ys = torch.arange(0, h * s, step=s, dtype=torch.float32)
        yy, xx = torch.meshgrid(ys, xs, indexing="ij")
        locs.append(torch.stack((xx.reshape(-1), yy.reshape(-1)), dim=1) + s // 2)
    return locs


def to_image_list(tensors, size_divisible=0):
    """structures/image_list.py:30-73 for a list of CHW tensors: zero-pad bottom / right to the largest size rounded up to
    a multiple of size_divisible; returns (batch [B,C,H,W], [(h, w) per image])."""
    max_size = [max(s) for s in zip(*[t.shape for t in tensors])]
    if size_divisible > 0:
        max_size[1] = int(math.ceil(max_size[1] / size_divisible) * size_divisible)
        max_size[2] = int(math.ceil(max_size[2] / size_divisible) * size_divisible)
    out = tensors[0].new_zeros((len(tensors),) + tuple(max_size))
    for t, o in zip(tensors, out):
        o[:t.shape[0], :t.shape[1], :t.shape[2]].copy_(t)
    return out, [tuple(t.shape[-2:]) for t in tensors]


def hot_path_forward(images, queries, sd, shots=1, query_sizes=None, emu=None):
    """generalized_rcnn.py:226-312 up to and including the FCOS head (eval or train; no BN/GN state differs).
    images [B,3,H,W]; queries [B*shots,3,h,w]; query_sizes: true (h, w) per query of a padded batch.
    emu: an Emulation (reduced-precision mode: rounds where the bf16 engines store); None = the fp32 restatement.
    Returns a dict of every intermediate."""
    B = images.shape[0]
    feats = backbone(images, sd, "backbone.", emu)
    qfeats = backbone(queries, sd, "supp_backbone.", emu)
    # to_image_list on a 4-D tensor: image_list.py:44-50; on a list: the true sizes (:69)
    q_sizes = [tuple(queries.shape[-2:])] * queries.shape[0] if query_sizes is None else list(query_sizes)
    pooled = query_pool(qfeats, q_sizes, B)
    combined = correlate(feats, pooled, emu)
    logits, bbox_reg, centerness = fcos_head(combined, sd, emu=emu)
    return dict(features=feats, query_features=qfeats, pooled=pooled, combined=combined,
                logits=logits, bbox_reg=bbox_reg, centerness=centerness)


# --------------------------------------------------------------------------------------------------------------------
# proposals: score, top-k, decode, clip, NMS
# --------------------------------------------------------------------------------------------------------------------
def nms(boxes, scores, thresh, cuda_semantics=False):
    """csrc/cpu/nms_cpu.cpp:6-65: greedy NMS, '+1' areas, suppress when IoU >= thresh (CPU); the CUDA kernel
    (csrc/cuda/nms.cu:60) suppresses when IoU > thresh — selectable.  Returns kept indices ascending (original
    order), int64.  float32 arithmetic throughout, like the reference."""
    boxes = np.asarray(boxes, dtype=np.float32).reshape(-1, 4)
    scores = np.asarray(scores, dtype=np.float32)
    n = boxes.shape[0]
    if n == 0:
        return np.zeros((0,), dtype=np.int64)
    x1, y1, x2, y2 = boxes[:, 0], boxes[:, 1], boxes[:, 2], boxes[:, 3]
    one = np.float32(1)
    areas = (x2 - x1 + one) * (y2 - y1 + one)
    order = np.argsort(-scores, kind="stable")
    suppressed = np.zeros(n, dtype=bool)
    thr = np.float32(thresh)
    for _i in range(n):
        i = order[_i]
        if suppressed[i]:
            continue
        rest = order[_i + 1:]
        xx1 = np.maximum(x1[i], x1[rest])
        yy1 = np.maximum(y1[i], y1[rest])
        xx2 = np.minimum(x2[i], x2[rest])
        yy2 = np.minimum(y2[i], y2[rest])
        w = np.maximum(np.float32(0), xx2 - xx1 + one)
        h = np.maximum(np.float32(0), yy2 - yy1 + one)
        inter = w * h
        ovr = inter / (areas[i] + areas[rest] - inter)
        hit = (ovr > thr) if cuda_semantics else (ovr >= thr)
        suppressed[rest[hit]] = True
    return np.nonzero(~suppressed)[0].astype(np.int64)


def fcos_postprocess(logits, bbox_reg, centerness, image_sizes, pre_nms_top_n=6000, post_nms_top_n=2000,
                     nms_thresh=0.8, cuda_nms=False):
    """modeling/rpn/fcos/inference.py:46-137 (forward_for_single_feature_map), :251-281 (forward) and :289-323
    (select_over_all_levels) for the non-RPN_ONLY factory branch (:337-351: pre_nms_thresh 0, num_classes 2,
    min_size 0).  Returns per image (boxes [K,4], scores [K]) ordered as the reference returns them."""
    level_hw = [tuple(t.shape[-2:]) for t in logits]
    locations = compute_locations(level_hw)
    N = logits[0].shape[0]
    per_image = [[] for _ in range(N)]
    for loc, cl, br, ct in zip(locations, logits, bbox_reg, centerness):
        n, c, h, w = cl.shape
        box_cls = cl.permute(0, 2, 3, 1).reshape(n, -1, 1).sigmoid()                      # :53-55
        reg = br.permute(0, 2, 3, 1).reshape(n, -1, 4)                                    # :67-68
        ctr = ct.permute(0, 2, 3, 1).reshape(n, -1).sigmoid()                             # :69-70
        cand = box_cls > 0                                                                # :72
        topn = cand.view(n, -1).sum(1).clamp(max=pre_nms_top_n)                           # :73-74
        box_cls = box_cls * ctr[:, :, None]                                               # :77
        for i in range(n):
            sc = box_cls[i][cand[i]]
            idx = cand[i].nonzero()[:, 0]
            rg, lc = reg[i][idx], loc[idx]
            if cand[i].sum().item() > topn[i].item():                                     # :97-102
                sc, ti = sc.topk(int(topn[i]), sorted=False)
                rg, lc = rg[ti], lc[ti]
            det = torch.stack([lc[:, 0] - rg[:, 0], lc[:, 1] - rg[:, 1],
                               lc[:, 0] + rg[:, 2], lc[:, 1] + rg[:, 3]], dim=1)          # :104-109
            ih, iw = image_sizes[i]
            det[:, 0].clamp_(min=0, max=iw - 1)                                           # bounding_box.py:214-219
            det[:, 1].clamp_(min=0, max=ih - 1)
            det[:, 2].clamp_(min=0, max=iw - 1)
            det[:, 3].clamp_(min=0, max=ih - 1)
            ws, hs = det[:, 2] - det[:, 0] + 1, det[:, 3] - det[:, 1] + 1                 # boxlist_ops.py:202-215
            keep = ((ws >= 0) & (hs >= 0)).nonzero().squeeze(1)
            per_image[i].append((det[keep], sc[keep]))
    results = []
    for i in range(N):
        boxes = torch.cat([b for b, _ in per_image[i]], dim=0)
        scores = torch.cat([s for _, s in per_image[i]], dim=0)
        keep = torch.from_numpy(nms(boxes.numpy(), scores.numpy(), nms_thresh, cuda_nms))  # boxlist_ops.py:10-34
        boxes, scores = boxes[keep], scores[keep]
        if len(keep) > post_nms_top_n > 0:                                                # inference.py:316-321
            _, si = torch.sort(scores, descending=True)
            boxes, scores = boxes[si[:post_nms_top_n]], scores[si[:post_nms_top_n]]
        results.append((boxes, scores))
    return results


def add_gt_proposals(proposals, gt_boxes_per_image):
    """modeling/rpn/fcos/inference.py:139-160 (training, non-RPN_ONLY, :276-279): cat_boxlist((proposal, gt_box)) with a
    dummy score of 1 for the ground-truth boxes.  proposals: list of (boxes [K,4], scores [K])."""
    out = []
    for (b, s), g in zip(proposals, gt_boxes_per_image):
        g = torch.as_tensor(g, dtype=b.dtype).reshape(-1, 4)
        out.append((torch.cat([b, g], dim=0), torch.cat([s, torch.ones(len(g), dtype=s.dtype)], dim=0)))
    return out


# --------------------------------------------------------------------------------------------------------------------
# FCOS loss (training)
# --------------------------------------------------------------------------------------------------------------------
def sigmoid_focal_loss_cpu_formula(logits, targets, gamma, alpha):
    """layers/sigmoid_focal_loss.py:42-54 — what the reference evaluates on CPU (log(p + 1e-6))."""
    num_classes = logits.shape[1]
    class_range = torch.arange(1, num_classes + 1, dtype=targets.dtype).unsqueeze(0)
    t = targets.unsqueeze(1)
    p = torch.sigmoid(logits)
    term1 = (1 - p) ** gamma * torch.log(p + 1e-6)
    term2 = p ** gamma * torch.log(1 - p + 1e-6)
    return -(t == class_range).float() * term1 * alpha - ((t != class_range) * (t >= 0)).float() * term2 * (1 - alpha)


def sigmoid_focal_loss_cuda_formula(logits, targets, gamma, alpha):
    """csrc/cuda/SigmoidFocalLoss_cuda.cu:21-58 — what the reference evaluates on a GPU and what the HIP kernel
    follows: log(max(p, FLT_MIN)) and the stable log(1-p) = -x*(x>=0) - log(1 + exp(x - 2x*(x>=0)))."""
    num_classes = logits.shape[1]
    d = torch.arange(1, num_classes + 1, dtype=targets.dtype).unsqueeze(0)
    t = targets.unsqueeze(1)
    c1 = (t == d).float()
    c2 = ((t >= 0) & (t != d)).float()
    x = logits
    p = 1.0 / (1.0 + torch.exp(-x))
    flt_min = torch.finfo(torch.float32).tiny
    term1 = torch.pow(1.0 - p, gamma) * torch.log(torch.clamp(p, min=flt_min))
    ge = (x >= 0).float()
    term2 = torch.pow(p, gamma) * (-1.0 * x * ge - torch.log(1.0 + torch.exp(x - 2.0 * x * ge)))
    return -c1 * term1 * alpha - c2 * term2 * (1.0 - alpha)


def giou_loss(pred, target, weight):
    """layers/iou_loss.py:10-49, loc_loss_type 'giou' ((I+1)/(U+1) smoothing :34)."""
    pl, pt, pr, pb = pred[:, 0], pred[:, 1], pred[:, 2], pred[:, 3]
    tl, tt, tr, tb = target[:, 0], target[:, 1], target[:, 2], target[:, 3]
    target_area = (tl + tr) * (tt + tb)
    pred_area = (pl + pr) * (pt + pb)
    w_int = torch.min(pl, tl) + torch.min(pr, tr)
    g_w = torch.max(pl, tl) + torch.max(pr, tr)
    h_int = torch.min(pb, tb) + torch.min(pt, tt)
    g_h = torch.max(pb, tb) + torch.max(pt, tt)
    ac = g_w * g_h + 1e-7
    a_int = w_int * h_int
    a_union = target_area + pred_area - a_int
    ious = (a_int + 1.0) / (a_union + 1.0)
    gious = ious - (ac - a_union) / ac
    losses = 1 - gious
    if weight is not None and weight.sum() > 0:
        return (losses * weight).sum() / weight.sum()
    return losses.mean()


def fcos_targets(locations, gt_boxes_per_image, radius=1.5):
    """modeling/rpn/fcos/loss.py:101-204 (prepare_targets + compute_targets_for_locations with CENTER_SAMPLE,
    get_sample_region :52-99).  locations: list of [n_l, 2]; gt: list of [n, 4] xyxy (labels all 1 after
    clean_targets fcos.py:133-143).  Returns level-first (labels [sum N*n_l] int64, reg_targets [.., 4])."""
    sizes = [[-1, 64], [64, 128], [128, 256], [256, 512], [512, INF]]
    npl = [len(l) for l in locations]
    soi = torch.cat([torch.tensor(sizes[l], dtype=torch.float32)[None].expand(n, -1) for l, n in enumerate(npl)], 0)
    pts = torch.cat(locations, dim=0)
    xs, ys = pts[:, 0], pts[:, 1]
    labels_all, reg_all = [], []
    for bboxes in gt_boxes_per_image:
        bboxes = torch.as_tensor(bboxes, dtype=torch.float32).reshape(-1, 4)
        area = (bboxes[:, 2] - bboxes[:, 0] + 1) * (bboxes[:, 3] - bboxes[:, 1] + 1)    # bounding_box.py:226-231
        l = xs[:, None] - bboxes[:, 0][None]
        t = ys[:, None] - bboxes[:, 1][None]
        r = bboxes[:, 2][None] - xs[:, None]
        b = bboxes[:, 3][None] - ys[:, None]
        reg = torch.stack([l, t, r, b], dim=2)
        # get_sample_region loss.py:52-99
        K, G = len(xs), bboxes.shape[0]
        gt = bboxes[None].expand(K, G, 4)
        cx = (gt[..., 0] + gt[..., 2]) / 2
        cy = (gt[..., 1] + gt[..., 3]) / 2
        if G == 0 or cx[..., 0].sum() == 0:                                               # :58-60
            inside = torch.zeros((K, G), dtype=torch.bool)
        else:
            cgt = torch.zeros_like(gt)
            beg = 0
            for lvl, n_p in enumerate(npl):
                end = beg + n_p
                st = FPN_STRIDES[lvl] * radius
                xmin, ymin = cx[beg:end] - st, cy[beg:end] - st
                xmax, ymax = cx[beg:end] + st, cy[beg:end] + st
                cgt[beg:end, :, 0] = torch.where(xmin > gt[beg:end, :, 0], xmin, gt[beg:end, :, 0])
                cgt[beg:end, :, 1] = torch.where(ymin > gt[beg:end, :, 1], ymin, gt[beg:end, :, 1])
                cgt[beg:end, :, 2] = torch.where(xmax > gt[beg:end, :, 2], gt[beg:end, :, 2], xmax)
                cgt[beg:end, :, 3] = torch.where(ymax > gt[beg:end, :, 3], gt[beg:end, :, 3], ymax)
                beg = end
            cb = torch.stack((xs[:, None] - cgt[..., 0], ys[:, None] - cgt[..., 1],
                              cgt[..., 2] - xs[:, None], cgt[..., 3] - ys[:, None]), -1)
            inside = cb.min(-1)[0] > 0
        mx = reg.max(dim=2)[0]
        cared = (mx >= soi[:, [0]]) & (mx <= soi[:, [1]])                                 # :180-184
        l2a = area[None].repeat(K, 1)
        l2a[inside == 0] = INF
        l2a[cared == 0] = INF
        min_area, inds = l2a.min(dim=1)                                                   # :186-196
        reg_i = reg[range(K), inds]
        lab = torch.ones(K, dtype=torch.int64)
        lab[min_area == INF] = 0
        labels_all.append(torch.split(lab, npl, dim=0))
        reg_all.append(torch.split(reg_i, npl, dim=0))
    labels = torch.cat([torch.cat([li[lvl] for li in labels_all], 0) for lvl in range(len(npl))], 0)
    regs = torch.cat([torch.cat([ri[lvl] for ri in reg_all], 0) for lvl in range(len(npl))], 0)
    return labels, regs


def centerness_targets(reg):
    """loss.py:206-211."""
    lr, tb = reg[:, [0, 2]], reg[:, [1, 3]]
    return torch.sqrt((lr.min(dim=-1)[0] / lr.max(dim=-1)[0]) * (tb.min(dim=-1)[0] / tb.max(dim=-1)[0]))


def fcos_loss(logits, bbox_reg, centerness, gt_boxes_per_image, gamma=2.0, alpha=0.25, focal="cuda"):
    """loss.py:213-276 (FCOSLossComputation.__call__).  focal = 'cuda' follows SigmoidFocalLoss_cuda.cu (what
    the HIP kernel implements), 'cpu' follows layers/sigmoid_focal_loss.py:42-54 (what the reference runs on CPU)."""
    N = logits[0].shape[0]
    locations = compute_locations([tuple(t.shape[-2:]) for t in logits])
    labels, reg_t = fcos_targets(locations, gt_boxes_per_image)
    cls_f = torch.cat([t.permute(0, 2, 3, 1).reshape(-1, 1) for t in logits], 0)
    reg_f = torch.cat([t.permute(0, 2, 3, 1).reshape(-1, 4) for t in bbox_reg], 0)
    ctr_f = torch.cat([t.permute(0, 2, 3, 1).reshape(-1) for t in centerness], 0)
    pos = torch.nonzero(labels > 0).squeeze(1)
    fl = sigmoid_focal_loss_cuda_formula if focal == "cuda" else sigmoid_focal_loss_cpu_formula
    cls_loss = fl(cls_f, labels.int(), gamma, alpha).sum() / (pos.numel() + N)            # :251-254
    reg_p, reg_tp, ctr_p = reg_f[pos], reg_t[pos], ctr_f[pos]
    if pos.numel() > 0:
        ct = centerness_targets(reg_tp)
        reg_loss = giou_loss(reg_p, reg_tp, ct)
        ctr_loss = F.binary_cross_entropy_with_logits(ctr_p, ct)
    else:
        reg_loss, ctr_loss = reg_p.sum(), ctr_p.sum()
    return cls_loss, reg_loss, ctr_loss, dict(labels=labels, reg_targets=reg_t, num_pos=int(pos.numel()))


def to_torch_state_dict(np_sd):
    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in np_sd.items()}
