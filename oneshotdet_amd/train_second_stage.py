"""The second stage inside the training step (TrainEngine mixin, second_stage=True): ROIBoxHead in training on the training
proposals — subsample, box head forward, cross-entropy + smooth-L1, the whole backward.  Reference: box_head.py:100-203."""

import torch

from . import ops, spec
from .ops import ACT_RELU


class SecondStage(object):
    def box_head_forward_backward(self, feats, qfeats, q_sizes, shots, proposals, gt_boxes, gt_count, keys=None,
                                  want_debug=False):
        """ROIBoxHead in training (box_head.py:100-203) on the training proposals (ground truth appended): subsample on the
        device, box head forward on the 128 sampled ROIs per image with the FIRST query of every image (the reference returns
        the losses from inside its loop over shots), cross-entropy + smooth-L1 with the weights 5 / 2.5, and the whole
        backward on the current stream: weight / bias / GroupNorm gradients into the flat buffer, the gradient w.r.t. the
        target FPN features as fp32 level maps and w.r.t. the query features' level.
        proposals = (boxes [N,P,4], scores, counts).  keys [N,P]: uniform randoms of the sampler (default: torch.rand).
        -> (losses [3] = (loss_classifier, loss_box_reg, sampled rows), gx: 5 fp32 maps, (query level, fp32 map [N,h,w,C]))"""
        from . import box_head as bh
        from . import model
        b, cv, dt = "roi_heads.box.", self.convs, self.dtype
        pb, _, pc = proposals
        n, P, _ = pb.shape
        S = spec.BOX_BATCH_PER_IMAGE
        if keys is None:
            keys = torch.rand((n, P), device=self.device, dtype=torch.float32)
        sb, sl, st, si, sc = ops.box_match_sample(pb, pc, gt_boxes, gt_count, keys, S, spec.BOX_POSITIVE_FRACTION,
                                                  spec.BOX_FG_IOU_THRESH, spec.BOX_REG_WEIGHTS)
        M = n * S
        slope, gr, eps = spec.BOX_LEAKY_SLOPE, spec.GN_GROUPS, spec.GN_EPS
        (g0, dg0), (b0, db0) = self.extra[b + "compress_dim_conv.1.weight"], self.extra[b + "compress_dim_conv.1.bias"]
        (g1, dg1), (b1, db1) = self.extra[b + "compress_dim_conv.4.weight"], self.extra[b + "compress_dim_conv.4.bias"]
        (g2, dg2), (b2, db2) = self.extra[b + "feature_aggreg.1.weight"], self.extra[b + "feature_aggreg.1.bias"]
        c0x, c0q, c3, ca = (cv[b + k] for k in ("compress_dim_conv.0x", "compress_dim_conv.0q", "compress_dim_conv.3",
                                                 "feature_aggreg.0"))
        fc6, fc7, cp = cv[b + "fc6"], cv[b + "fc7"], cv[b + "pred"]
        # ---- forward
        qf1 = qfeats if shots == 1 else [q[::shots].contiguous() for q in qfeats]
        qs1 = [q_sizes[i * shots] for i in range(n)]
        uniform = len(set(qs1)) == 1
        q = bh.run_query_roi(qf1, qs1[0] if uniform else qs1, dt)                                # [N,7,7,C]
        qh = ops.conv2d(q, c0q.pc)                                                               # W_q q + b
        x = ops.roi_pool_levels(feats, spec.POOLER_SCALES, sb, sc, spec.BOX_POOL, spec.POOLER_SAMPLING_RATIO)
        u0 = ops.conv2d(x, c0x.pc)
        t0 = ops.groupnorm_act_rois(u0, g0, b0, gr, eps, slope, addend=qh, rois_per_add=S, add_stride=1, add_offset=0)
        u1 = ops.conv2d(t0, c3.pc)
        t1 = ops.groupnorm_act_rois(u1, g1, b1, gr, eps, slope)
        u2 = ops.conv2d(t1, ca.pc, pad=1)
        t2 = ops.groupnorm_act_rois(u2, g2, b2, gr, eps, slope)
        t2f = t2.view(M, 1, 1, -1)
        f6 = ops.conv2d(t2f, fc6.pc, act=ACT_RELU)
        f7 = ops.conv2d(f6, fc7.pc, act=ACT_RELU)
        pred = ops.conv2d(f7, cp.pc)
        losses, d_pred = ops.box_loss(pred, sl, st, sc, n, S, spec.BOX_LOSS_WEIGHTS[0], spec.BOX_LOSS_WEIGHTS[1],
                                      grad_stride=cp.pd.cin_k)
        # ---- backward (inline on this stream: M = 1024 ROIs)

        def wg(c, xin, dy, pad=0):
            ops.conv2d_wgrad(xin, dy, c.gw, c.r, c.s, 1, pad, c.cout, db=c.gb if c.has_bias else None)

        def dg(c, dy, mask=None):
            return ops.conv2d(dy, c.pd, pad=c.r - 1 - (c.r // 2), mask=mask)
        d_pred = d_pred.view(M, 1, 1, -1)
        wg(cp, f7, d_pred)
        d_f7 = dg(cp, d_pred, mask=f7)
        wg(fc7, f6, d_f7)
        d_f6 = dg(fc7, d_f7, mask=f6)
        wg(fc6, t2f, d_f6)
        d_t2 = dg(fc6, d_f6).view(t2.shape)
        d_u2 = ops.groupnorm_act_rois_bwd(u2, g2, b2, d_t2, dg2, db2, gr, eps, slope)
        wg(ca, t1, d_u2, pad=1)
        d_t1 = dg(ca, d_u2)
        d_u1 = ops.groupnorm_act_rois_bwd(u1, g1, b1, d_t1, dg1, db1, gr, eps, slope)
        wg(c3, t0, d_u1)
        d_t0 = dg(c3, d_u1)
        d_u0 = ops.groupnorm_act_rois_bwd(u0, g0, b0, d_t0, dg0, db0, gr, eps, slope, addend=qh, rois_per_add=S, add_stride=1,
                                          add_offset=0)
        wg(c0x, x, d_u0)
        d_x = dg(c0x, d_u0)
        d_qh = ops.rois_sum(d_u0, n, S)                     # the query half was added to every ROI of its image
        wg(c0q, q, d_qh)
        d_q = dg(c0q, d_qh)
        gx = ops.roi_pool_levels_bwd([(f.shape[1], f.shape[2]) for f in feats], spec.POOLER_SCALES, sb, sc, d_x, spec.BOX_POOL,
                                     spec.POOLER_SAMPLING_RATIO)
        if uniform:
            lvl = bh.query_level(*qs1[0])
            rois = model.whole_image_rois(qs1, self.device)
            gq = ops.roi_align_bwd(d_q.float(), rois, qf1[lvl].shape, spec.POOLER_SCALES[lvl], spec.BOX_POOL, spec.BOX_POOL,
                                   spec.POOLER_SAMPLING_RATIO)
            gqs = [(lvl, gq)]
        else:                                                # padded query batch: every whole-image box picks its own level
            boxes = model.whole_image_rois(qs1, self.device)[:, 1:].reshape(n, 1, 4).contiguous()
            maps = ops.roi_pool_levels_bwd([(f.shape[1], f.shape[2]) for f in qf1], spec.POOLER_SCALES, boxes, None, d_q,
                                           spec.BOX_POOL, spec.POOLER_SAMPLING_RATIO)
            gqs = list(enumerate(maps))
        self._keep.append((sb, sl, st, si, sc, q, qh, x, u0, t0, u1, t1, u2, t2, f6, f7, pred, d_pred, d_f7, d_f6, d_t2, d_u2,
                           d_t1, d_u1, d_t0, d_u0, d_x, d_qh, d_q, keys))
        if want_debug:
            self.last_box = dict(boxes=sb, labels=sl, targets=st, index=si, counts=sc, pred=pred)
        return losses, gx, gqs
