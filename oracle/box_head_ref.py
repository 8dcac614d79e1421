"""ORACLE — test infrastructure only.  CPU restatement of the reference's second-stage few-shot ROI box head
(SURVEY.md §8f #1), inference path, for the config of record (SECOND_STAGE_METHOD 'concat', no negative support,
SECOND_STAGE_CLS_LOSS 'ce_loss', 2 classes, FPN2ROIFeatureExtractor + FPNPredictor).

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this module; the product
(`oneshotdet_amd/`) never does.  Paths below are relative to /root/reference/maskrcnn_benchmark/.  Pinned against the
REAL reference by tests/golden/make_golden.py (`gen_box_case`: same synthetic weights and inputs through
`model.roi_heads.box` and `model.supproi_pooling`, agreement asserted, fixtures `tests/golden/box_*.npz` written) and
re-checked everywhere by tests/test_oracle_golden.py.

ATen arithmetic (conv2d, group_norm, leaky_relu, linear, softmax, exp, log2) is outside /root/reference and pinned by the
fixtures only.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from .hotpath_ref import POOLER_SCALES, nms

POOL = 7
SAMPLING_RATIO = 2
REG_WEIGHTS = (10.0, 10.0, 5.0, 5.0)
BBOX_XFORM_CLIP = math.log(1000.0 / 16)          # modeling/box_coder.py:19


def map_levels(boxes, k_min=3, k_max=7, s0=224, lvl0=4, eps=1e-6):
    """modeling/poolers.py:33-42 (LevelMapper.__call__) with BoxList.area's '+1' (structures/bounding_box.py:226-236);
    Pooler sets k_min/k_max = -log2(scales[0]) / -log2(scales[-1]) (poolers.py:73-75).  boxes [R,4] fp32 -> int64 [R]
    level index into the 5 feature maps."""
    area = (boxes[:, 2] - boxes[:, 0] + 1) * (boxes[:, 3] - boxes[:, 1] + 1)
    s = torch.sqrt(area)
    lv = torch.floor(lvl0 + torch.log2(s / s0 + eps))
    lv = torch.clamp(lv, min=k_min, max=k_max)
    return lv.to(torch.int64) - k_min


def roi_align_vec(inp, rois, spatial_scale, ph, pw, sampling_ratio):
    """csrc/cpu/ROIAlign_cpu.cpp:114-219 for sampling_ratio > 0, vectorised over ROIs (same fp32 operation order as
    oracle.hotpath_ref.roi_align, which walks one ROI at a time; tests compare the two).
    inp [B,C,H,W] fp32, rois [R,5] -> [R,C,ph,pw]."""
    assert sampling_ratio > 0
    B, C, H, W = inp.shape
    R = rois.shape[0]
    if R == 0:
        return inp.new_zeros((0, C, ph, pw))
    r = rois.detach().to(torch.float32)
    bi = r[:, 0].to(torch.int64)
    scale = torch.tensor(spatial_scale, dtype=torch.float32)
    rsw, rsh, rew, reh = r[:, 1] * scale, r[:, 2] * scale, r[:, 3] * scale, r[:, 4] * scale
    roi_w = torch.clamp(rew - rsw, min=1.0)
    roi_h = torch.clamp(reh - rsh, min=1.0)
    bin_h, bin_w = roi_h / float(ph), roi_w / float(pw)
    g = sampling_ratio
    flat = inp.permute(0, 2, 3, 1).reshape(B * H * W, C)            # gather rows of C channels
    out = inp.new_zeros((R, ph, pw, C))
    i_idx = torch.arange(ph, dtype=torch.float32).view(1, ph, 1)
    j_idx = torch.arange(pw, dtype=torch.float32).view(1, 1, pw)
    for iy in range(g):
        yy = rsh.view(R, 1, 1) + i_idx * bin_h.view(R, 1, 1) + (float(iy) + 0.5) * bin_h.view(R, 1, 1) / float(g)
        for ix in range(g):
            xx = rsw.view(R, 1, 1) + j_idx * bin_w.view(R, 1, 1) + (float(ix) + 0.5) * bin_w.view(R, 1, 1) / float(g)
            y = yy.expand(R, ph, pw)
            x = xx.expand(R, ph, pw)
            outside = (y < -1.0) | (y > H) | (x < -1.0) | (x > W)    # ROIAlign_cpu.cpp:59-69
            y = torch.clamp(y, min=0.0)
            x = torch.clamp(x, min=0.0)
            y_low, x_low = y.to(torch.int64), x.to(torch.int64)
            top = y_low >= H - 1
            y_low = torch.where(top, torch.full_like(y_low, H - 1), y_low)
            y_high = torch.where(top, y_low, y_low + 1)
            y = torch.where(top, y_low.to(torch.float32), y)
            right = x_low >= W - 1
            x_low = torch.where(right, torch.full_like(x_low, W - 1), x_low)
            x_high = torch.where(right, x_low, x_low + 1)
            x = torch.where(right, x_low.to(torch.float32), x)
            ly, lx = y - y_low.to(torch.float32), x - x_low.to(torch.float32)
            hy, hx = 1.0 - ly, 1.0 - lx
            base = (bi.view(R, 1, 1) * H).expand(R, ph, pw)
            def rows(yi, xi):
                return flat[((base + yi) * W + xi).reshape(-1)].view(R, ph, pw, C)
            term = ((hy * hx).unsqueeze(-1) * rows(y_low, x_low) + (hy * lx).unsqueeze(-1) * rows(y_low, x_high)
                    + (ly * hx).unsqueeze(-1) * rows(y_high, x_low) + (ly * lx).unsqueeze(-1) * rows(y_high, x_high))
            out = out + torch.where(outside.unsqueeze(-1), torch.zeros_like(term), term)
    out = out / float(g * g)
    return out.permute(0, 3, 1, 2).contiguous()


def pooler(feats, boxes_per_image):
    """modeling/poolers.py:93-124 (Pooler.forward, 5 levels): every ROI is pooled from the level LevelMapper picks.
    feats 5 x [B,C,H,W]; boxes_per_image list of [R,4] (equal R: poolers.py:80) -> [B, R, C, 7, 7]."""
    B = feats[0].shape[0]
    counts = {len(b) for b in boxes_per_image}
    assert len(counts) == 1, counts
    rois = torch.cat([torch.cat([torch.full((len(b), 1), float(i)), b.to(torch.float32)], dim=1)
                      for i, b in enumerate(boxes_per_image)], dim=0)
    levels = map_levels(rois[:, 1:])
    out = feats[0].new_zeros((rois.shape[0], feats[0].shape[1], POOL, POOL))
    for lvl, (f, scale) in enumerate(zip(feats, POOLER_SCALES)):
        idx = torch.nonzero(levels == lvl).squeeze(1)
        if idx.numel():
            out[idx] = roi_align_vec(f, rois[idx], scale, POOL, POOL, SAMPLING_RATIO)
    return out.view(B, rois.shape[0] // B, out.shape[1], POOL, POOL)


def query_roi_features(query_feats, query_sizes):
    """generalized_rcnn.py:257,290 (supproi_pooling on one whole-image box per query image, the (h, w)-as-(x2, y2)
    quirk included) + supproi_pooling.py:62-66 -> [B*S, 1, C, 7, 7]."""
    boxes = [torch.tensor([[0.0, 0.0, float(h), float(w)]]) for (h, w) in query_sizes]
    return pooler(query_feats, boxes)


def _gn_leaky(x, sd, p, groups=32):
    return F.leaky_relu(F.group_norm(x, groups, sd[p + ".weight"], sd[p + ".bias"], 1e-5), 0.2)


def box_head_logits(feats, query_feats, proposals, query_sizes, sd, prefix="roi_heads.box."):
    """modeling/roi_heads/box_head/box_head.py:81-257 (ROIBoxHead.forward, eval, 'concat' without negative support).
    proposals: list of [R,4] boxes per image.  Returns class_logits [B*R,2], box_regression [B*R,8] (after the
    per-class arg-max over shots of :239-252 when there is more than one query per image) and the pooled ROI features."""
    x = pooler(feats, proposals)                                             # feature_extractor, :106
    B, R, C, Hh, Ww = x.shape
    supp = query_roi_features(query_feats, query_sizes)                     # [B*S,1,C,7,7]
    supp = supp.view(B, -1, C, Hh, Ww)                                       # :115
    all_logits, all_reg = [], []
    for s in range(supp.shape[1]):                                           # :122
        e = supp[:, [s]].expand_as(x).contiguous().view(-1, C, Hh, Ww)       # :126
        y = torch.cat((x.view(-1, C, Hh, Ww), e), dim=1)                     # :146
        y = F.conv2d(y, sd[prefix + "compress_dim_conv.0.weight"], sd[prefix + "compress_dim_conv.0.bias"])
        y = _gn_leaky(y, sd, prefix + "compress_dim_conv.1")
        y = F.conv2d(y, sd[prefix + "compress_dim_conv.3.weight"], sd[prefix + "compress_dim_conv.3.bias"])
        y = _gn_leaky(y, sd, prefix + "compress_dim_conv.4")                 # :148
        y = F.conv2d(y, sd[prefix + "feature_aggreg.0.weight"], sd[prefix + "feature_aggreg.0.bias"], padding=1)
        y = _gn_leaky(y, sd, prefix + "feature_aggreg.1")                    # :150
        y = y.view(y.size(0), -1)
        y = F.relu(F.linear(y, sd[prefix + "fc6.weight"], sd[prefix + "fc6.bias"]))      # :152
        y = F.relu(F.linear(y, sd[prefix + "fc7.weight"], sd[prefix + "fc7.bias"]))      # :153
        all_logits.append(F.linear(y, sd[prefix + "predictor.cls_score.weight"], sd[prefix + "predictor.cls_score.bias"]))
        all_reg.append(F.linear(y, sd[prefix + "predictor.bbox_pred.weight"], sd[prefix + "predictor.bbox_pred.bias"]))
    if len(all_logits) > 1:                                                   # :239-252
        tl, tr = torch.stack(all_logits, 0), torch.stack(all_reg, 0)
        idx = torch.argmax(tl, dim=0)                                         # [B*R, 2]
        logits = torch.gather(tl, 0, idx.unsqueeze(0))[0]
        bidx = idx[:, :, None].expand(-1, -1, 4).reshape(idx.shape[0], -1)
        reg = torch.gather(tr, 0, bidx.unsqueeze(0))[0]
    else:
        logits, reg = all_logits[0], all_reg[0]
    return logits, reg, x


def decode_boxes(rel_codes, boxes, weights=REG_WEIGHTS):
    """modeling/box_coder.py:50-95 (BoxCoder.decode)."""
    boxes = boxes.to(rel_codes.dtype)
    widths = boxes[:, 2] - boxes[:, 0] + 1
    heights = boxes[:, 3] - boxes[:, 1] + 1
    ctr_x = boxes[:, 0] + 0.5 * widths
    ctr_y = boxes[:, 1] + 0.5 * heights
    wx, wy, ww, wh = weights
    dx, dy = rel_codes[:, 0::4] / wx, rel_codes[:, 1::4] / wy
    dw = torch.clamp(rel_codes[:, 2::4] / ww, max=BBOX_XFORM_CLIP)
    dh = torch.clamp(rel_codes[:, 3::4] / wh, max=BBOX_XFORM_CLIP)
    pcx = dx * widths[:, None] + ctr_x[:, None]
    pcy = dy * heights[:, None] + ctr_y[:, None]
    pw_ = torch.exp(dw) * widths[:, None]
    ph_ = torch.exp(dh) * heights[:, None]
    out = torch.zeros_like(rel_codes)
    out[:, 0::4] = pcx - 0.5 * pw_
    out[:, 1::4] = pcy - 0.5 * ph_
    out[:, 2::4] = pcx + 0.5 * pw_ - 1
    out[:, 3::4] = pcy + 0.5 * ph_ - 1
    return out


def box_postprocess(logits, reg, proposals, image_sizes, score_thresh=0.0, nms_thresh=0.5, detections_per_img=2000,
                    cuda_nms=False):
    """modeling/roi_heads/box_head/inference.py:46-166 (PostProcessor.forward + filter_results), 'ce_loss' branch:
    softmax over the 2 logits, decode all 2x4 deltas, clip, per class j >= 1: score > thresh, NMS, keep at most
    detections_per_img by score.  image_sizes: (h, w) per image.  Returns per image (boxes [K,4], scores [K]) in the
    order the reference returns them (ascending proposal index: nms_cpu.cpp:64 / nms.cu:127-130)."""
    reg = reg[:, :8]
    prob = F.softmax(logits, -1)[:, :2]
    counts = [len(p) for p in proposals]
    dec = decode_boxes(reg.view(sum(counts), -1), torch.cat(proposals, dim=0))
    results = []
    for pr, bx, (ih, iw) in zip(prob.split(counts, 0), dec.split(counts, 0), image_sizes):
        bx = bx.reshape(-1, 4).clone()
        bx[:, 0].clamp_(min=0, max=iw - 1)                                    # bounding_box.py:214-219
        bx[:, 1].clamp_(min=0, max=ih - 1)
        bx[:, 2].clamp_(min=0, max=iw - 1)
        bx[:, 3].clamp_(min=0, max=ih - 1)
        bx = bx.reshape(-1, 8)
        inds = torch.nonzero(pr[:, 1] > score_thresh).squeeze(1)              # :136-141, j = 1 only (2 classes)
        sc, bj = pr[inds, 1], bx[inds, 4:8]
        keep = torch.from_numpy(nms(bj.numpy(), sc.numpy(), nms_thresh, cuda_nms))
        bj, sc = bj[keep], sc[keep]
        if len(keep) > detections_per_img > 0:                                # :158-163
            _, si = torch.sort(sc, descending=True)
            bj, sc = bj[si[:detections_per_img]], sc[si[:detections_per_img]]
        results.append((bj, sc))
    return results


def box_head_forward(feats, query_feats, proposals, image_sizes, query_sizes, sd, cuda_nms=False):
    """Second stage end to end: target FPN features (NOT the correlated ones: generalized_rcnn.py:317), query FPN
    features, first-stage proposals -> detections."""
    logits, reg, pooled = box_head_logits(feats, query_feats, proposals, query_sizes, sd)
    det = box_postprocess(logits, reg, proposals, image_sizes, cuda_nms=cuda_nms)
    return dict(logits=logits, box_regression=reg, pooled=pooled, detections=det)
