"""Host side of the siamese-FCOS hot path on MI355X: weight packing and kernel sequencing.

Mirrors, function for function, the reference call stack of SURVEY.md §3.1:
  GeneralizedRCNN.forward (modeling/detector/generalized_rcnn.py:226-312)
    backbone / supp_backbone  = Sequential(ResNet body, FPN)   (modeling/backbone/backbone.py:51-72)
    supp_pooling + batch_pooling + correlation                   (generalized_rcnn.py:297-311)
    rpn = FCOSModule: FCOSHead + FCOSPostProcessor               (modeling/rpn/fcos/fcos.py, inference.py)
All tensors between kernels are NHWC; state_dict names and OIHW shapes are the reference's (oneshotdet_amd/spec.py).
"""
import os

import torch

from . import streams as _streams

from . import ops, spec
from .ops import ACT_EXP_SCALE, ACT_NONE, ACT_RELU, RES_DOWN2X, RES_SAME, RES_UP2X


LOCKSTEP = os.environ.get("OSD_LOCKSTEP", "0") != "0"            # inference engine: both backbones per launch (A/B)
TOWERS_MERGED = os.environ.get("OSD_TOWERS_MERGED", "0") != "0"  # inference engine: both towers per launch (A/B)
SKIP_UNUSED_C2 = os.environ.get("OSD_FULL_C2", "0") == "0"     # A/B switch: compute all of layer1's last block anyway


def _bn(sd, p):
    return (sd[p + ".weight"], sd[p + ".bias"], sd[p + ".running_mean"], sd[p + ".running_var"])


RES_DOWN2X_OK = os.environ.get("OSD_NO_RES_DOWN2X", "0") == "0"      # A/B switch: copy the identity's even pixels instead (the form until round 6)
FPN_OUT_GROUPED = os.environ.get("OSD_NO_FPN_GROUPED", "0") == "0"       # the FPN's P3 + P4 output convs as one launch
FUSE_DOWNSAMPLE = os.environ.get("OSD_NO_FUSE_DS", "0") == "0"      # A/B switch: conv3 + downsample of a stage's first block as one GEMM


def pack_conv3_downsample(sd, p, dtype):
    """The first block of a stage ends in relu(bn3(conv3(out)) + bn_d(downsample(x))) (resnet.py:295-315).  Both are 1x1
    convs onto the same pixels, so they are ONE GEMM over the concatenated input channels [out | x]: FrozenBN folded into
    the rows of either part (batch_norm.py:19-24, no eps), the shifts added.  The 4 x-wide downsample output is then never
    written and never read back as a residual (osd_conv2d_fwd with a second source)."""
    def fold(conv, bn):
        g, b, mean, var = (t.float() for t in _bn(sd, p + bn))
        scale = g * var.rsqrt()
        return sd[p + conv + ".weight"].float() * scale.view(-1, 1, 1, 1), b - mean * scale
    w3, b3 = fold("conv3", "bn3")
    wd, bd = fold("downsample.0", "downsample.1")
    return ops.pack_conv(torch.cat([w3, wd], 1).contiguous(), bias=(b3 + bd).contiguous(), dtype=dtype)


class BackboneWeights(object):
    """Packed weights of one ResNet-50-FPN (resnet.py:80-145, fpn.py:28-41,82-94)."""

    def __init__(self, sd, prefix, dtype):
        b = prefix + "body."
        self.stem = ops.pack_conv(sd[b + "stem.conv1.weight"], bn=_bn(sd, b + "stem.bn1"), dtype=dtype, stem=True)
        self.blocks = []
        for si, nblocks in enumerate(spec.STAGE_BLOCKS):
            for bi in range(nblocks):
                p = "%slayer%d.%d." % (b, si + 1, bi)
                blk = {"stride": 2 if (bi == 0 and si > 0) else 1, "ds": None}
                if (p + "downsample.0.weight") in sd:
                    blk["ds"] = ops.pack_conv(sd[p + "downsample.0.weight"], bn=_bn(sd, p + "downsample.1"), dtype=dtype)
                    # bf16 only: the exact-fp32 engines stay on the reference's operation order (the fused sum rounds once
                    # instead of twice — enough to flip a ReLU that sits on zero, which the fp32 gradient tests would see)
                    blk["c3ds"] = pack_conv3_downsample(sd, p, dtype) if (FUSE_DOWNSAMPLE and dtype == torch.bfloat16) else None
                for i in (1, 2, 3):
                    blk["c%d" % i] = ops.pack_conv(sd["%sconv%d.weight" % (p, i)], bn=_bn(sd, "%sbn%d" % (p, i)),
                                                   dtype=dtype)
                blk["last_of_stage"] = bi == nblocks - 1
                self.blocks.append(blk)
        f = prefix + "fpn."
        self.fpn = {}
        for name in ("fpn_inner2", "fpn_inner3", "fpn_inner4", "fpn_layer2", "fpn_layer3", "fpn_layer4",
                     "top_blocks.p6", "top_blocks.p7"):
            self.fpn[name] = ops.pack_conv(sd[f + name + ".weight"], bias=sd[f + name + ".bias"], dtype=dtype)


class HeadWeights(object):
    """Packed FCOSHead (fcos.py:27-81).  cls_logits and centerness both read the cls tower (fcos.py:91-92), so they
    are fused into one 2-output conv (stored as 4 channels: logit, centerness, 0, 0)."""

    def __init__(self, sd, dtype, prefix="rpn.head."):
        self.towers = {}
        for tower in ("cls_tower", "bbox_tower"):
            layers = []
            for i in range(spec.NUM_CONVS):
                conv = ops.pack_conv(sd["%s%s.%d.weight" % (prefix, tower, 3 * i)],
                                     bias=sd["%s%s.%d.bias" % (prefix, tower, 3 * i)], dtype=dtype)
                layers.append((conv, sd["%s%s.%d.weight" % (prefix, tower, 3 * i + 1)].float().contiguous(),
                               sd["%s%s.%d.bias" % (prefix, tower, 3 * i + 1)].float().contiguous()))
            self.towers[tower] = layers
        w = torch.cat([sd[prefix + "cls_logits.weight"], sd[prefix + "centerness.weight"]], 0)
        b = torch.cat([sd[prefix + "cls_logits.bias"], sd[prefix + "centerness.bias"]], 0)
        self.pred_cls_ctr = ops.pack_conv(w, bias=b, dtype=dtype)
        self.pred_box = ops.pack_conv(sd[prefix + "bbox_pred.weight"], bias=sd[prefix + "bbox_pred.bias"], dtype=dtype)
        # Scale modules (layers/scale.py): the scalar is folded into the exp epilogue; one host read at pack time
        self.scales = [float(sd["%sscales.%d.scale" % (prefix, i)].item()) for i in range(5)]
        self.scales_dev = torch.cat([sd["%sscales.%d.scale" % (prefix, i)].reshape(1) for i in range(5)]).float().contiguous()


def run_backbone(wts, images, dtype, return_body=False):
    """images NCHW fp32 -> [P3, P4, P5, P6, P7] NHWC.  resnet.py:138-145,295-315,332-337; fpn.py:43-75,95-99."""
    x, (ho, wo) = ops.stem_input(images, dtype)
    x = ops.conv2d(x, wts.stem, act=ACT_RELU, out_hw=(ho, wo))
    x = ops.maxpool3x3s2(x)
    feats = []
    halved = False
    for bi, blk in enumerate(wts.blocks):
        s = blk["stride"]
        if s == 2 and halved:
            s, halved = 1, False
        # C2 is read only by layer2.0's stride-2 1x1 convs (the FPN skips it): the last block of layer1 computes just the
        # even pixels (3x3 at stride 2, then the 1x1 + residual on the quarter-size map); return_body keeps the full map
        quarter = SKIP_UNUSED_C2 and not return_body and bi == spec.STAGE_BLOCKS[0] - 1
        if blk.get("c3ds") is not None:       # first block of a stage: conv3 + downsample as one GEMM over [out | x]
            out = ops.conv2d(x, blk["c1"], stride=s, act=ACT_RELU)
            out = ops.conv2d(out, blk["c2"], pad=1, act=ACT_RELU)
            x = ops.conv2d(out, blk["c3ds"], act=ACT_RELU, x2=x, x2_stride=s)
            if blk["last_of_stage"]:
                feats.append(x)
            continue
        identity = x if blk["ds"] is None else ops.conv2d(x, blk["ds"], stride=s)
        out = ops.conv2d(x, blk["c1"], stride=s, act=ACT_RELU)
        rmode = RES_SAME
        if quarter:
            out = ops.conv2d(out, blk["c2"], stride=2, pad=1, act=ACT_RELU)
            if RES_DOWN2X_OK:        # the 1x1's epilogue reads the identity at (2 ho, 2 wo) itself (round 6: no strided copy of a 210 MB map)
                rmode = RES_DOWN2X
            else:
                identity = identity[:, ::2, ::2].contiguous()
            halved = True
        else:
            out = ops.conv2d(out, blk["c2"], pad=1, act=ACT_RELU)
        x = ops.conv2d(out, blk["c3"], act=ACT_RELU, res=identity, res_mode=rmode)
        if blk["last_of_stage"]:
            feats.append(x)
    c3, c4, c5 = feats[1], feats[2], feats[3]
    f = wts.fpn
    inner4 = ops.conv2d(c5, f["fpn_inner4"])
    p5 = ops.conv2d(inner4, f["fpn_layer4"], pad=1)
    inner3 = ops.conv2d(c4, f["fpn_inner3"], res=inner4, res_mode=RES_UP2X)
    inner2 = ops.conv2d(c3, f["fpn_inner2"], res=inner3, res_mode=RES_UP2X)
    if FPN_OUT_GROUPED:      # the P3 and P4 output convs as one launch: 400 + 100 pixel tiles = 1.95 rounds of 256 CUs instead of 1.56 and 0.4
        p3, p4 = ops.conv2d_multi([inner2, inner3], [f["fpn_layer2"], f["fpn_layer3"]], pad=1)
    else:
        p4 = ops.conv2d(inner3, f["fpn_layer3"], pad=1)
        p3 = ops.conv2d(inner2, f["fpn_layer2"], pad=1)
    p6 = ops.conv2d(p5, f["top_blocks.p6"], stride=2, pad=1)
    p7 = ops.conv2d(p6, f["top_blocks.p7"], stride=2, pad=1, relu_in=True)
    out = [p3, p4, p5, p6, p7]
    return (out, feats) if return_body else out


def run_backbones(wt, wq, images, queries, dtype):
    """Target and query backbone (BackboneWeights wt / wq: generalized_rcnn.py:270-272, separately parameterised, same
    graph) in LOCKSTEP: every layer is one osd_conv2d_fwd_multi launch over (target, query), so the query branch's
    latency-sized launches ride in the tail of the target's.  -> ([P3..P7] target, [P3..P7] query), NHWC."""
    xs = []
    for wts, im in ((wt, images), (wq, queries)):
        x, (ho, wo) = ops.stem_input(im, dtype)
        x = ops.conv2d(x, wts.stem, act=ACT_RELU, out_hw=(ho, wo))
        xs.append(ops.maxpool3x3s2(x))
    feats = []
    halved = False
    for bi, (bt, bq) in enumerate(zip(wt.blocks, wq.blocks)):
        s = bt["stride"]
        if s == 2 and halved:
            s, halved = 1, False
        quarter = SKIP_UNUSED_C2 and bi == spec.STAGE_BLOCKS[0] - 1         # see run_backbone
        identity = xs if bt["ds"] is None else ops.conv2d_multi(xs, [bt["ds"], bq["ds"]], stride=s)
        out = ops.conv2d_multi(xs, [bt["c1"], bq["c1"]], stride=s, act=ACT_RELU)
        rmode = RES_SAME
        if quarter:
            out = ops.conv2d_multi(out, [bt["c2"], bq["c2"]], stride=2, pad=1, act=ACT_RELU)
            if RES_DOWN2X_OK and all(t.shape[1] % 2 == 0 and t.shape[2] % 2 == 0 for t in identity):
                rmode = RES_DOWN2X
            else:
                identity = [t[:, ::2, ::2].contiguous() for t in identity]
            halved = True
        else:
            out = ops.conv2d_multi(out, [bt["c2"], bq["c2"]], pad=1, act=ACT_RELU)
        xs = ops.conv2d_multi(out, [bt["c3"], bq["c3"]], act=ACT_RELU, residuals=identity, res_mode=rmode)
        if bt["last_of_stage"]:
            feats.append(xs)
    c3, c4, c5 = feats[1], feats[2], feats[3]

    def f(name):
        return [wt.fpn[name], wq.fpn[name]]
    inner4 = ops.conv2d_multi(c5, f("fpn_inner4"))
    p5 = ops.conv2d_multi(inner4, f("fpn_layer4"), pad=1)
    inner3 = ops.conv2d_multi(c4, f("fpn_inner3"), residuals=inner4, res_mode=RES_UP2X)
    inner2 = ops.conv2d_multi(c3, f("fpn_inner2"), residuals=inner3, res_mode=RES_UP2X)
    if FPN_OUT_GROUPED:
        p34 = ops.conv2d_multi(inner2 + inner3, f("fpn_layer2") + f("fpn_layer3"), pad=1)
        p3, p4 = p34[:2], p34[2:]
    else:
        p4 = ops.conv2d_multi(inner3, f("fpn_layer3"), pad=1)
        p3 = ops.conv2d_multi(inner2, f("fpn_layer2"), pad=1)
    p6 = ops.conv2d_multi(p5, f("top_blocks.p6"), stride=2, pad=1)
    p7 = [ops.conv2d(p6[j], f("top_blocks.p7")[j], stride=2, pad=1, relu_in=True) for j in (0, 1)]   # relu prologue: tiny
    return [[p3[j], p4[j], p5[j], p6[j], p7[j]] for j in (0, 1)]


_ROI_CACHE = {}


def whole_image_rois(q_sizes, device):
    """[i, 0, 0, h, w] per query — the reference's quirk: the box is built from image_sizes = (h, w) but consumed as
    (x1, y1, x2, y2) (generalized_rcnn.py:257).  Cached per device so no host->device copy happens in steady state
    (and none inside a hipGraph capture)."""
    key = (tuple(q_sizes), str(device))
    if key not in _ROI_CACHE:
        _ROI_CACHE[key] = torch.tensor([[float(i), 0.0, 0.0, float(h), float(w)] for i, (h, w) in enumerate(q_sizes)],
                                       dtype=torch.float32, device=device)
    return _ROI_CACHE[key]


_SIZE_CACHE = {}


def image_sizes_dev(image_sizes, device):
    """[N,2] fp32 (height, width) per image on the device, cached like the ROI table (no copy in steady state)."""
    key = (tuple(tuple(int(v) for v in s) for s in image_sizes), str(device))
    if key not in _SIZE_CACHE:
        _SIZE_CACHE[key] = torch.tensor([[float(h), float(w)] for h, w in key[0]], dtype=torch.float32, device=device)
    return _SIZE_CACHE[key]


def run_query_pool(qfeats, q_sizes, batch):
    """SuppAlignLayer (generalized_rcnn.py:20-52) + batch_pooling (:100-104) -> 5 x [B, C] fp32."""
    rois = whole_image_rois(q_sizes, qfeats[0].device)
    if ops.QUERY_POOL_LEVELS and len(qfeats) <= 8:
        return ops.query_pool_levels(qfeats, rois, spec.POOLER_SCALES, batch, spec.POOLER_SAMPLING_RATIO)
    pooled = []
    for feat, scale in zip(qfeats, spec.POOLER_SCALES):
        v = ops.roi_align(feat, rois, scale, 1, 1, spec.POOLER_SAMPLING_RATIO)
        pooled.append(ops.shot_mean(v.view(v.shape[0], -1), batch))
    return pooled


def run_correlate(feats, pooled):
    """generalized_rcnn.py:307-311: all five levels in one launch."""
    return ops.correlate_levels(feats, pooled)


def run_head_tower(hw, feats, tower):
    """One tower + its prediction conv over all levels (fcos.py:89-97): per layer ONE grouped conv launch over the
    levels (they share the weights), then GroupNorm+ReLU of all levels in two launches."""
    t = list(feats)
    for conv, gamma, beta in hw.towers[tower]:
        u = ops.conv2d_grouped(t, conv, pad=1)
        t, _ = ops.groupnorm_relu_levels(u, gamma, beta, spec.GN_GROUPS, spec.GN_EPS)
    if tower == "cls_tower":
        return ops.conv2d_grouped(t, hw.pred_cls_ctr, pad=1)
    return ops.conv2d_grouped(t, hw.pred_box, pad=1, act=ACT_EXP_SCALE,
                              act_scale_devs=[hw.scales_dev[l:l + 1] for l in range(len(t))])


def run_head(hw, feats, streams=None):
    """FCOSHead.forward (fcos.py:83-99).  Per level returns (cls_ctr [N,H,W,4] = (logit, centerness, 0, 0),
    reg [N,H,W,4] = exp(scale_l * bbox_pred)).  The two towers are independent: with `streams` the bbox tower runs on a
    side stream beside the cls tower, so the HBM-bound GroupNorm passes of one tower overlap the MFMA-bound convs of the
    other and the small levels' launch-latency-bound kernels fill the CUs the P3 GEMMs leave idle.  OSD_TOWERS_MERGED=1
    runs each layer of both towers as ONE launch instead (fewer launches, less total kernel time, but measured SLOWER end
    to end: 5.5 vs 4.7 ms per forward step — DESIGN 6b)."""
    if TOWERS_MERGED:
        return run_head_merged(hw, feats)
    if not streams:
        return list(zip(run_head_tower(hw, feats, "cls_tower"), run_head_tower(hw, feats, "bbox_tower")))
    main = _streams.current()
    streams[0].wait_stream(main)
    with _streams.on(streams[0]):
        box_out = run_head_tower(hw, feats, "bbox_tower")
    cls_out = run_head_tower(hw, feats, "cls_tower")
    main.wait_stream(streams[0])
    return list(zip(cls_out, box_out))


def run_head_merged(hw, feats):
    """Both towers per launch: each layer is ONE launch over both towers and all levels (10 pairs, level-major so the
    tuner's large / small split keeps P3 and P4 together), then the GroupNorm+ReLU of a tower's levels in two launches."""
    nl = len(feats)
    towers = ("cls_tower", "bbox_tower")
    t = {tw: list(feats) for tw in towers}
    for i in range(spec.NUM_CONVS):
        xs = [t[tw][l] for l in range(nl) for tw in towers]
        pcs = [hw.towers[tw][i][0] for l in range(nl) for tw in towers]
        us = ops.conv2d_multi(xs, pcs, pad=1)
        for k, tw in enumerate(towers):
            _, gamma, beta = hw.towers[tw][i]
            t[tw], _ = ops.groupnorm_relu_levels(us[k::2], gamma, beta, spec.GN_GROUPS, spec.GN_EPS)
    cls_out = ops.conv2d_grouped(t["cls_tower"], hw.pred_cls_ctr, pad=1)
    box_out = ops.conv2d_grouped(t["bbox_tower"], hw.pred_box, pad=1, act=ACT_EXP_SCALE,
                                 act_scale_devs=[hw.scales_dev[l:l + 1] for l in range(nl)])
    return list(zip(cls_out, box_out))


class ProposalDepth(object):
    """Lagged feedback for the proposal pipeline (osd_proposals_sort_nms_hint): how deep into the score order the greedy
    NMS had to read in recent calls decides how many candidates the next call sorts exactly in its first phase.  The
    depth travels device -> pinned host memory by a non-blocking copy behind the call; the host looks at it only once
    that copy's event has completed (no synchronisation: a step or two of lag), so the pipeline itself never waits."""

    def __init__(self):
        self.hint, self._pending = 0, None

    def before(self, n, device):
        if self._pending is not None and self._pending[1].query():
            host = self._pending[0]
            d = int(host.max())
            # a quarter of headroom over the deepest image, never below the default (0 = let the library decide)
            self.hint = d + d // 4 + 320
            self._pending = None
        return torch.empty((n,), device=device, dtype=torch.int32)

    def after(self, depth_dev):
        if self._pending is None:
            host = torch.empty(depth_dev.shape, dtype=torch.int32, pin_memory=True)
            host.copy_(depth_dev, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            self._pending = (host, ev, depth_dev)


def run_proposals(head_out, img_h, img_w, pre_nms_top_n, post_nms_top_n, nms_thresh, cuda_nms=True, workspace=None,
                  image_sizes=None, depth=None):
    """FCOSPostProcessor.forward (fcos/inference.py:251-323).  Boxes are clipped to (img_h, img_w) (the 4-D tensor path of
    to_image_list, structures/image_list.py:44-50) or, for a padded batch, to every image's own image_sizes[i] = (h, w)
    (:52-70).  Everything stays on the device; returns boxes [N, post, 4], scores [N, post] (descending), counts [N]."""
    n = head_out[0][0].shape[0]
    dev = head_out[0][0].device
    sizes = [(c.shape[1], c.shape[2]) for c, _ in head_out]
    total = sum(h * w for h, w in sizes)
    scores = torch.empty((n, total), device=dev, dtype=torch.float32)
    boxes = torch.empty((n, total, 4), device=dev, dtype=torch.float32)
    off = 0
    offs = []
    hw_dev = None if image_sizes is None else image_sizes_dev(image_sizes, dev)
    for (cls_ctr, reg), stride in zip(head_out, spec.FPN_STRIDES):
        ops.fcos_score_decode(cls_ctr, reg, scores, boxes, stride, off, img_h, img_w, hw_dev)
        offs.append(off)
        off += cls_ctr.shape[1] * cls_ctr.shape[2]
    levels = [(lo, h * w) for (h, w), lo in zip(sizes, offs)]
    max_count = sum(min(c, pre_nms_top_n) for _, c in levels)
    if os.environ.get("OSD_PROPOSALS_FULL_SORT"):      # A/B: rank all candidates (two separate calls)
        bs, ss, idx, cnt = ops.rank_sort_gather(scores, boxes, max_count, levels, pre_nms_top_n)
        ob, os_, op, oc = ops.nms_sorted(bs, ss, cnt, nms_thresh, post_nms_top_n, cuda_semantics=cuda_nms,
                                         workspace=workspace)
        return ob, os_, oc
    if depth is None:
        return ops.proposals_sort_nms(scores, boxes, max_count, levels, pre_nms_top_n, nms_thresh, post_nms_top_n,
                                      cuda_semantics=cuda_nms)
    d = depth.before(n, dev)
    out = ops.proposals_sort_nms(scores, boxes, max_count, levels, pre_nms_top_n, nms_thresh, post_nms_top_n,
                                 cuda_semantics=cuda_nms, head_hint=depth.hint, depth_out=d)
    depth.after(d)
    return out


class HotPathEngine(object):
    """Packed weights + forward of the hot path.  state_dict: reference-named fp32 tensors (any device; moved to
    `device`).  dtype: torch.float32 (exact-fp32 MFMA) or torch.bfloat16 (bf16 MFMA, fp32 accumulate)."""

    def __init__(self, state_dict, dtype=torch.float32, device="cuda"):
        if not torch.cuda.is_available():
            raise ops._lib.OsdError("HotPathEngine needs an MI355X: no GPU visible and there is no CPU fallback")
        ops._lib.load()
        self.device = torch.device(device)
        self.dtype = dtype
        self.sd = {k: torch.as_tensor(v).to(self.device, torch.float32) for k, v in state_dict.items()}
        missing = [k for k in spec.hot_path_shapes() if k not in self.sd]
        if missing:
            raise KeyError("state_dict is missing hot-path keys, e.g. %s" % missing[:3])
        self.repack()

    def repack(self):
        self.backbone = BackboneWeights(self.sd, "backbone.", self.dtype)
        self.supp_backbone = BackboneWeights(self.sd, "supp_backbone.", self.dtype)
        self.head = HeadWeights(self.sd, self.dtype)
        # second stage: packed when the state_dict carries the reference's roi_heads.box.* entries
        self.box_head = None
        if all(k in self.sd for k in spec.box_head_shapes()):
            from .box_head import BoxHeadWeights
            self.box_head = BoxHeadWeights(self.sd, self.dtype)

    def tune(self, images, queries, second_stage=False):
        """Pick, by measurement on this device, the conv algorithm (kernel generation, ring depth, tile) for every
        distinct conv shape of this input geometry (cached in ops.ALGO_CACHE; a few ms per shape, done once)."""
        with ops.tuning():
            self.detect(images, queries, concurrent=False, second_stage=second_stage)
        torch.cuda.synchronize()

    def side_streams(self):
        if getattr(self, "_streams", None) is None:
            self._streams = [torch.cuda.Stream(device=self.device) for _ in range(3)]
        return self._streams

    def forward_features(self, images, queries, concurrent=True, query_sizes=None):
        """Target backbone on the current stream; the (independent, tiny) query backbone + pooling on a side stream
        (concurrent=False: one after the other on the current stream; OSD_LOCKSTEP=1: both backbones per launch, A/B).
        query_sizes: true (h, w) of every query of a padded batch (default: the tensor's size)."""
        batch = images.shape[0]
        q_sizes = [tuple(queries.shape[-2:])] * queries.shape[0] if query_sizes is None else list(query_sizes)
        if concurrent and LOCKSTEP:
            feats, qfeats = run_backbones(self.backbone, self.supp_backbone, images, queries, self.dtype)
            pooled = run_query_pool(qfeats, q_sizes, batch)
        elif concurrent:
            main, side = _streams.current(), self.side_streams()[0]
            side.wait_stream(main)
            with _streams.on(side):
                qfeats = run_backbone(self.supp_backbone, queries, self.dtype)
                pooled = run_query_pool(qfeats, q_sizes, batch)
            feats = run_backbone(self.backbone, images, self.dtype)
            main.wait_stream(side)
        else:
            feats = run_backbone(self.backbone, images, self.dtype)
            qfeats = run_backbone(self.supp_backbone, queries, self.dtype)
            pooled = run_query_pool(qfeats, q_sizes, batch)
        combined = run_correlate(feats, pooled)
        return feats, qfeats, pooled, combined

    def forward(self, images, queries, concurrent=True, query_sizes=None):
        """images [B,3,H,W], queries [B*S,3,h,w] NCHW fp32 on the device -> dict of NHWC intermediates."""
        feats, qfeats, pooled, combined = self.forward_features(images, queries, concurrent, query_sizes)
        head = run_head(self.head, combined, self.side_streams() if concurrent else None)
        return dict(features=feats, query_features=qfeats, pooled=pooled, combined=combined, head=head)

    def detect(self, images, queries, training=False, cuda_nms=True, concurrent=True, second_stage=False):
        """GeneralizedRCNN.forward in eval mode (generalized_rcnn.py:226-332): first stage -> out["proposals"] =
        (boxes [N,P,4], scores [N,P], counts [N]); with second_stage also the few-shot ROI box head ->
        out["detections"] = dict(boxes [N,K,4], scores [N,K] descending, counts [N])."""
        from .layers import ImageList
        image_sizes = query_sizes = None
        if isinstance(images, ImageList):       # padded batch (R0): clip to every image's own size
            images, image_sizes = images.tensors, images.image_sizes
        elif isinstance(images, ops.PackedImages):
            image_sizes = images.image_sizes
        if isinstance(queries, ops.PackedImages):
            query_sizes = queries.image_sizes
        if isinstance(queries, ImageList):
            queries, query_sizes = queries.tensors, queries.image_sizes
        out = self.forward(images, queries, concurrent, query_sizes)
        h, w = images.shape[-2:]
        pre = spec.PRE_NMS_TOP_N_TRAIN if training else spec.PRE_NMS_TOP_N_TEST
        post = spec.POST_NMS_TOP_N_TRAIN if training else spec.POST_NMS_TOP_N_TEST
        out["proposals"] = run_proposals(out["head"], h, w, pre, post, spec.NMS_THRESH, cuda_nms, image_sizes=image_sizes)
        if second_stage:
            q_sizes = query_sizes if query_sizes is not None else tuple(queries.shape[-2:])
            out["detections"] = self.box_detect(out["features"], out["query_features"], q_sizes,
                                                out["proposals"][0], out["proposals"][2], h, w,
                                                shots=queries.shape[0] // images.shape[0], cuda_nms=cuda_nms,
                                                image_sizes=image_sizes)
        return out

    def box_detect(self, feats, qfeats, q_size, boxes, counts, img_h, img_w, shots=1, cuda_nms=True, want_raw=False,
                   image_sizes=None):
        """roi_heads.box on given proposals (boxes [N,R,4] fp32, counts [N] int32 or None).  q_size: (h, w) of all queries
        or a list with every query's true size; image_sizes: true (h, w) per image of a padded batch."""
        if self.box_head is None:
            raise KeyError("state_dict has no roi_heads.box.* entries: the second stage was not packed")
        from .box_head import run_box_head
        hw_dev = None if image_sizes is None else image_sizes_dev(image_sizes, boxes.device)
        return run_box_head(self.box_head, feats, qfeats, q_size, boxes, counts, img_h, img_w, shots=shots,
                            cuda_nms=cuda_nms, want_raw=want_raw, img_hw=hw_dev)


class GraphedDetect(object):
    """The whole hot-path forward captured once into a hipGraph (static shapes, static HBM buffers) and replayed:
    ~250 kernel launches per step cost one graph launch on the host, and the multi-stream fork/join of
    HotPathEngine.forward becomes parallel branches of the graph."""

    def __init__(self, engine, images, queries, warmup=2, **kw):
        self.engine = engine
        self.images = images.clone()
        self.queries = queries.clone()
        engine.tune(self.images, self.queries, second_stage=bool(kw.get("second_stage")))
        side = torch.cuda.Stream(device=engine.device)
        side.wait_stream(_streams.current())
        with _streams.on(side):
            for _ in range(warmup):
                engine.detect(self.images, self.queries, **kw)
        _streams.current().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.out = engine.detect(self.images, self.queries, **kw)

    def __call__(self, images=None, queries=None):
        if images is not None:
            self.images.copy_(images, non_blocking=True)
        if queries is not None:
            self.queries.copy_(queries, non_blocking=True)
        self.graph.replay()
        return self.out
