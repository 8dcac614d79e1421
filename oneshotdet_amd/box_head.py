"""Second-stage few-shot ROI box head on MI355X, inference path.

Mirrors ROIBoxHead.forward (modeling/roi_heads/box_head/box_head.py:81-259) for the config of record
(SECOND_STAGE_METHOD 'concat', no negative support, 'ce_loss'):
  feature_extractor = Pooler 7x7 over P3..P7 of the TARGET backbone (not the correlated maps: generalized_rcnn.py:317)
  supproi_pooling   = the same Pooler on one whole-image box per query (generalized_rcnn.py:257,290)
  cat(x, query) -> conv1x1 512->512, GN, LeakyReLU -> conv1x1 512->256, GN, LeakyReLU -> conv3x3 256->128, GN, LeakyReLU
  -> fc6 6272->1024, ReLU -> fc7, ReLU -> cls_score (2) | bbox_pred (8) -> softmax, decode, clip, NMS 0.5.
MI355X design: the first 1x1 conv is split at the concatenation, conv(cat(x, q)) = W_x x + (W_q q + b).  The query half
is one 7x7 map per (image, shot) instead of one per ROI; it is added inside the GroupNorm kernel that follows, so the
concatenated [R,7,7,512] tensor never exists, the conv does half the FLOPs, and with several shots the ROI half is
computed once.  The Linear layers are 1x1 convs on the implicit-GEMM kernels (fc6 reads the [R,7,7,128] maps as
[R,1,1,6272] with its weight columns permuted to (h, w, c) order at pack time).
"""
import numpy as np
import torch

from . import ops, spec
from .ops import ACT_NONE, ACT_RELU, PackedConv


class BoxHeadWeights(object):
    """Packed `roi_heads.box.*` (spec.box_head_shapes)."""

    def __init__(self, sd, dtype, prefix="roi_heads.box."):
        c = spec.FPN_OUT
        w0, b0 = sd[prefix + "compress_dim_conv.0.weight"], sd[prefix + "compress_dim_conv.0.bias"]
        self.conv0_x = ops.pack_conv(w0[:, :c].contiguous(), bias=None, dtype=dtype)          # ROI half, no bias
        self.conv0_q = ops.pack_conv(w0[:, c:].contiguous(), bias=b0, dtype=dtype)            # query half + bias
        self.conv3 = ops.pack_conv(sd[prefix + "compress_dim_conv.3.weight"], bias=sd[prefix + "compress_dim_conv.3.bias"],
                                   dtype=dtype)
        self.aggreg = ops.pack_conv(sd[prefix + "feature_aggreg.0.weight"], bias=sd[prefix + "feature_aggreg.0.bias"],
                                    dtype=dtype)
        self.gn = [tuple(sd[prefix + name + "." + leaf].float().contiguous() for leaf in ("weight", "bias"))
                   for name in ("compress_dim_conv.1", "compress_dim_conv.4", "feature_aggreg.1")]
        p, mid = spec.BOX_POOL, c // 2
        # fc6 columns are (c, h, w) in the reference (x.view(N, -1) of NCHW, box_head.py:151); as an OIHW conv weight the
        # packer writes them in (h, w, c) order = the flattening of this build's NHWC ROI maps
        fc6 = ops.pack_conv(sd[prefix + "fc6.weight"].reshape(spec.BOX_MLP_DIM, mid, p, p), bias=sd[prefix + "fc6.bias"],
                            dtype=dtype)
        self.fc6 = PackedConv(fc6.w.view(fc6.w_rows, 1, 1, p * p * fc6.cin_k), fc6.bias, fc6.cout, fc6.cout_store,
                              fc6.w_rows, p * p * fc6.cin_k, 1, 1)
        self.fc7 = ops.pack_conv(sd[prefix + "fc7.weight"][:, :, None, None], bias=sd[prefix + "fc7.bias"], dtype=dtype)
        wp = torch.cat([sd[prefix + "predictor.cls_score.weight"], sd[prefix + "predictor.bbox_pred.weight"]], 0)
        bp = torch.cat([sd[prefix + "predictor.cls_score.bias"], sd[prefix + "predictor.bbox_pred.bias"]], 0)
        self.pred = ops.pack_conv(wp[:, :, None, None].contiguous(), bias=bp, dtype=dtype)  # cols 0..1 cls, 2..9 deltas


def query_level(qh, qw):
    """LevelMapper for the whole-image query box [0, 0, h, w] (generalized_rcnn.py:257), fp32 like the reference."""
    f = np.float32
    s = np.sqrt((f(qh) + f(1)) * (f(qw) + f(1)), dtype=np.float32)
    lv = np.floor(f(spec.LEVEL_MAP_LEVEL) + np.log2(s / f(spec.LEVEL_MAP_SCALE) + f(spec.LEVEL_MAP_EPS), dtype=np.float32))
    return int(min(max(lv, 3), 3 + len(spec.POOLER_SCALES) - 1)) - 3


def run_query_roi(qfeats, q_size, dtype):
    """supproi_pooling: [B*S, 7, 7, C] in the activation dtype.  q_size: (h, w) shared by all queries, or one per query
    (padded batch): each whole-image box goes through the level-routed Pooler like any other ROI."""
    from .model import whole_image_rois
    n = qfeats[0].shape[0]
    sizes = [tuple(q_size)] * n if isinstance(q_size[0], int) else [tuple(s) for s in q_size]
    if len(set(sizes)) == 1:
        lvl = query_level(*sizes[0])
        rois = whole_image_rois(sizes, qfeats[0].device)
        v = ops.roi_align(qfeats[lvl], rois, spec.POOLER_SCALES[lvl], spec.BOX_POOL, spec.BOX_POOL, spec.POOLER_SAMPLING_RATIO)
        return v if dtype == torch.float32 else ops.cast_f32(v, dtype)
    boxes = whole_image_rois(sizes, qfeats[0].device)[:, 1:].reshape(n, 1, 4).contiguous()      # [0, 0, h, w] per query
    return ops.roi_pool_levels(qfeats, spec.POOLER_SCALES, boxes, None, spec.BOX_POOL, spec.POOLER_SAMPLING_RATIO)


def run_box_head(bw, feats, qfeats, q_size, boxes, counts, img_h, img_w, shots=1, cuda_nms=True, want_raw=False, img_hw=None):
    """feats / qfeats: NHWC P3..P7 of the target / query backbone; boxes [N,R,4], counts [N] = first-stage proposals.
    Returns dict(boxes [N,K,4], scores [N,K] descending, counts [N]) (+ raw logits / deltas / pooled maps)."""
    n, r, _ = boxes.shape
    dtype = feats[0].dtype
    q = run_query_roi(qfeats, q_size, dtype)                                           # [N*S,7,7,C]
    assert q.shape[0] == n * shots
    q_half = ops.conv2d(q, bw.conv0_q)                                                 # [N*S,7,7,512]  W_q q + b
    x = ops.roi_pool_levels(feats, spec.POOLER_SCALES, boxes, counts, spec.BOX_POOL, spec.POOLER_SAMPLING_RATIO)
    u0 = ops.conv2d(x, bw.conv0_x)                                                     # [N*R,7,7,512]  W_x x
    preds = torch.empty((shots, n * r, bw.pred.cout_store), device=boxes.device, dtype=dtype)
    for s in range(shots):                                                             # box_head.py:122
        (g0, b0), (g1, b1), (g2, b2) = bw.gn
        t = ops.groupnorm_act_rois(u0, g0, b0, spec.GN_GROUPS, spec.GN_EPS, spec.BOX_LEAKY_SLOPE, addend=q_half,
                                   rois_per_add=r, add_stride=shots, add_offset=s)
        t = ops.conv2d(t, bw.conv3)
        t = ops.groupnorm_act_rois(t, g1, b1, spec.GN_GROUPS, spec.GN_EPS, spec.BOX_LEAKY_SLOPE, out=t)
        t = ops.conv2d(t, bw.aggreg, pad=1)
        t = ops.groupnorm_act_rois(t, g2, b2, spec.GN_GROUPS, spec.GN_EPS, spec.BOX_LEAKY_SLOPE, out=t)
        t = ops.conv2d(t.view(n * r, 1, 1, -1), bw.fc6, act=ACT_RELU)
        t = ops.conv2d(t, bw.fc7, act=ACT_RELU)
        ops.conv2d(t, bw.pred, act=ACT_NONE, out=preds[s].view(n * r, 1, 1, -1))
    dec = ops.box_decode(preds, boxes, counts, spec.BOX_REG_WEIGHTS, img_h, img_w, spec.BOX_SCORE_THRESH, want_raw=want_raw,
                         img_hw=img_hw)
    scores, dboxes = dec[0], dec[1]
    bs, ss, _, cnt = ops.rank_sort_gather(scores, dboxes, r)
    keep = min(spec.BOX_DETECTIONS_PER_IMG, r)
    ob, os_, _, oc = ops.nms_sorted(bs, ss, cnt, spec.BOX_NMS_THRESH, keep, cuda_semantics=cuda_nms)
    out = dict(boxes=ob, scores=os_, counts=oc)
    if want_raw:
        out.update(logits=dec[2], box_regression=dec[3], pooled=x, query_roi=q)
    return out
