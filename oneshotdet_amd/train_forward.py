"""Forward half of the training step (TrainEngine mixin): both R-50-FPN backbones with saved activations, the FCOS head's towers
layer by layer (conv launches over all FPN levels + level-grouped GroupNorm), the FCOS loss with the gradients w.r.t. the
prediction convs.  Reference: generalized_rcnn.py:226-312, fcos.py:83-99, fcos/loss.py:213-276."""

import torch

from . import streams

from . import model as _model
from . import ops, spec
from .ops import ACT_EXP_SCALE, ACT_RELU, RES_DOWN2X, RES_SAME, RES_UP2X

SIZE_RANGES = ((-1.0, 64.0), (64.0, 128.0), (128.0, 256.0), (256.0, 512.0), (512.0, float(spec.INF)))


class ForwardPass(object):
    def backbones_forward(self, images, queries, after_frozen=None):
        return self._backbones_forward(self.BBS, (images, queries), after_frozen)

    def _backbones_forward(self, bbs, inputs, after_frozen=None):
        """Both R-50-FPN backbones (generalized_rcnn.py:270-272: separately parameterised, same graph) in LOCKSTEP: every
        layer is ONE osd_conv2d_fwd_multi launch over (target, query), so the query branch's latency-sized launches (M = 8
        .. 8192 pixels) ride in the tail of the target's instead of costing ~110 launches of their own per step.
        after_frozen: called once the stems and layer1 (frozen: resnet.py:127-136) have been enqueued, before the first
        layer that reads trainable weights.  Returns ([feats_target, feats_query], [ctx_target, ctx_query])."""
        cv, dt = self.convs, self.dtype
        nb = range(len(bbs))

        def pcs(name):
            return [cv[bb + name].pc for bb in bbs]
        xs = []
        for bb, im in zip(bbs, inputs):
            x, (ho, wo) = ops.stem_input(im, dt)
            x = ops.conv2d(x, cv[bb + "body.stem.conv1"].pc, act=ACT_RELU, out_hw=(ho, wo))
            xs.append(ops.maxpool3x3s2(x))
        blocks, stage_out = [[] for _ in nb], [[] for _ in nb]
        halved = False
        for si, nblocks in enumerate(spec.STAGE_BLOCKS):
            for bi in range(nblocks):
                p = "body.layer%d.%d." % (si + 1, bi)
                s = 2 if (bi == 0 and si > 0) else 1
                if s == 2 and halved:        # the stride already happened in the producer (see below)
                    s, halved = 1, False
                has_ds = (bbs[0] + p + "downsample.0") in cv
                # C2 (layer1's output) is read by nothing but layer2.0's two stride-2 1x1 convs (the FPN skips it, fpn.py:33,
                # backbone.py:59): only its even pixels are ever used, so the last block of the FROZEN layer1 computes
                # just those — its 3x3 at stride 2, its 1x1 + residual on the quarter-size map — and layer2.0 reads
                # them at stride 1.  Same values, 3/4 of two convs and of a 210 MB tensor gone.
                quarter = self.skip_unused_c2 and si == 0 and bi == nblocks - 1 and len(spec.STAGE_BLOCKS) > 1
                if si == 0 and has_ds and self._fused_l1:
                    # frozen layer1.0 (no backward through it): conv3 + downsample as one GEMM over [conv2 output | block
                    # input] (model.pack_conv3_downsample): the 4x-wide downsample map is never written nor re-read
                    o1 = ops.conv2d_multi(xs, pcs(p + "conv1"), stride=s, act=ACT_RELU)
                    o2 = ops.conv2d_multi(o1, pcs(p + "conv2"), pad=1, act=ACT_RELU)
                    xs = [ops.conv2d(o2[j], self._fused_l1[bbs[j]], act=ACT_RELU, x2=xs[j], x2_stride=s) for j in nb]
                    continue
                identity = ops.conv2d_multi(xs, pcs(p + "downsample.0"), stride=s) if has_ds else xs
                o1 = ops.conv2d_multi(xs, pcs(p + "conv1"), stride=s, act=ACT_RELU)
                rmode = RES_SAME
                if quarter:
                    o2 = ops.conv2d_multi(o1, pcs(p + "conv2"), stride=2, pad=1, act=ACT_RELU)
                    if _model.RES_DOWN2X_OK and all(t.shape[1] % 2 == 0 and t.shape[2] % 2 == 0 for t in identity):
                        rmode = RES_DOWN2X       # conv3's epilogue reads the identity at (2 ho, 2 wo): no strided copy of the 210 MB map
                    else:
                        identity = [t[:, ::2, ::2].contiguous() for t in identity]
                    halved = True
                else:
                    o2 = ops.conv2d_multi(o1, pcs(p + "conv2"), pad=1, act=ACT_RELU)
                y = ops.conv2d_multi(o2, pcs(p + "conv3"), act=ACT_RELU, residuals=identity, res_mode=rmode)
                if si >= 1:
                    for j in nb:
                        blocks[j].append(dict(p=p, s=s, ds=has_ds, x=xs[j], o1=o1[j], o2=o2[j], y=y[j],
                                              first=(si == 1 and bi == 0)))
                xs = y
            for j in nb:
                stage_out[j].append(xs[j])
            if si == 0 and after_frozen is not None:
                after_frozen()
        c3, c4, c5 = ([so[i] for so in stage_out] for i in (1, 2, 3))
        f = "fpn."
        inner4 = ops.conv2d_multi(c5, pcs(f + "fpn_inner4"))
        p5 = ops.conv2d_multi(inner4, pcs(f + "fpn_layer4"), pad=1)
        inner3 = ops.conv2d_multi(c4, pcs(f + "fpn_inner3"), residuals=inner4, res_mode=RES_UP2X)
        inner2 = ops.conv2d_multi(c3, pcs(f + "fpn_inner2"), residuals=inner3, res_mode=RES_UP2X)
        if self.fpn_out_grouped:
            # the P3 and P4 output convs (fpn.py:66-75: same 3x3 256 -> 256 geometry, their own weights) as ONE launch: alone, P3's
            # 400 pixel tiles are 1.56 rounds of the 256 CUs and P4's 100 fill 40 % of one; together 500 tiles are 1.95 rounds —
            # the tower launch's shape (the grouped-launch tuner may still cut it where that measures faster)
            p34 = ops.conv2d_multi(inner2 + inner3, pcs(f + "fpn_layer2") + pcs(f + "fpn_layer3"), pad=1)
            p3, p4 = p34[:len(inner2)], p34[len(inner2):]
        else:
            p4 = ops.conv2d_multi(inner3, pcs(f + "fpn_layer3"), pad=1)
            p3 = ops.conv2d_multi(inner2, pcs(f + "fpn_layer2"), pad=1)
        p6 = ops.conv2d_multi(p5, pcs(f + "top_blocks.p6"), stride=2, pad=1)
        p6r = [ops.add_mask(t, None, t) for t in p6]        # relu(P6), materialised: the P7 weight gradient reads it
        p7 = ops.conv2d_multi(p6r, pcs(f + "top_blocks.p7"), stride=2, pad=1)
        feats, ctxs = [], []
        for j in nb:
            feats.append([p3[j], p4[j], p5[j], p6[j], p7[j]])
            ctxs.append(dict(bb=bbs[j], blocks=blocks[j], c3=c3[j], c4=c4[j], c5=c5[j], inner4=inner4[j], inner3=inner3[j],
                             inner2=inner2[j], p5=p5[j], p6=p6[j], p6r=p6r[j]))
        return feats, ctxs

    def head_forward(self, feats):
        """FCOSHead.forward (fcos.py:83-99).  Layer by layer, BOTH towers over all five levels = ONE conv launch per layer
        (10 pairs: they share the geometry, each tower brings its own weights; level-major order so that the tuner's
        large / small split keeps P3 and P4 of both towers together), then GroupNorm+ReLU of a tower's five levels in two
        launches.  ctx[tower] = ([per layer: (inputs per level, conv outputs per level, ab)], last activations)."""
        if self.towers_merged:
            outs, ctx = self._towers_forward(feats, self.TOWERS)
            return list(zip(outs["cls_tower"], outs["bbox_tower"])), ctx
        # one stream per tower: the HBM-bound GroupNorm passes of one tower run beside the MFMA-bound convs of the other
        main = streams.current()
        side = self.s1 if self.s1 is not None else main
        if self.split_levels and len(feats) == 5 and None not in (self.s1, self.wstream, self.wstream2):
            # ... and one CHAIN per level group: the levels of a tower never meet before the loss, so P5-P7 (34 pixel tiles,
            # latency-sized launches: 27 us per layer) run as their own conv -> GroupNorm chain on the weight-gradient streams
            # (idle during the forward pass) inside the HBM-bound GroupNorm windows of the P3+P4 chain, instead of in line
            big, small = [0, 1], [2, 3, 4]
            for st in (self.s1, self.wstream, self.wstream2):
                st.wait_stream(main)
            with streams.on(self.wstream):
                ocs, ccs = self._towers_forward(feats, ("cls_tower",), small)
            with streams.on(self.wstream2):
                obs, cbs = self._towers_forward(feats, ("bbox_tower",), small)
            with streams.on(self.s1):
                obb, cbb = self._towers_forward(feats, ("bbox_tower",), big)
            ocb, ccb = self._towers_forward(feats, ("cls_tower",), big)
            for st in (self.s1, self.wstream, self.wstream2):
                main.wait_stream(st)

            def merge(cb_, cs_, tw):
                lb, tb = cb_[tw]
                ls, ts = cs_[tw]
                return ([(a[0] + b[0], a[1] + b[1], [a[2], b[2]]) for a, b in zip(lb, ls)], tb + ts)
            ctx = {"cls_tower": merge(ccb, ccs, "cls_tower"), "bbox_tower": merge(cbb, cbs, "bbox_tower")}
            return list(zip(ocb["cls_tower"] + ocs["cls_tower"], obb["bbox_tower"] + obs["bbox_tower"])), ctx
        side.wait_stream(main)
        with streams.on(side):
            ob, cb = self._towers_forward(feats, ("bbox_tower",))
        oc, cc = self._towers_forward(feats, ("cls_tower",))
        main.wait_stream(side)
        cc.update(cb)
        return list(zip(oc["cls_tower"], ob["bbox_tower"])), cc

    def _towers_forward(self, feats, towers, lv=None):
        cv = self.convs
        h = "rpn.head."
        scales = self.extra[h + "scales"][0]
        lv = list(range(len(feats))) if lv is None else list(lv)        # FPN levels handled by this call
        feats = [feats[l] for l in lv]
        nl, nt = len(feats), len(towers)
        t = {tw: list(feats) for tw in towers}
        layers = {tw: [] for tw in towers}
        # GroupNorm forward statistics (sum, sum of squares per image and group) are gathered by the tower conv's epilogue where
        # the kernel can (ops.gn_bwd_fusable: the large levels); the GroupNorm then skips its statistics pass for those levels.
        # The sums are atomic adds: not in ordered mode
        n_img, c_gn = feats[0].shape[0], cv["%s%s.0" % (h, towers[0])].pc.cout_store
        nf = 0
        if self.fuse_gn_fwd:
            while nf < nl and all(ops.gn_bwd_fusable(feats[nf], cv["%s%s.0" % (h, tw)].pc, 1, 1) for tw in towers):
                nf += 1
        if nf > 0:
            per = nl * n_img * ops.GN_SPLITS * spec.GN_GROUPS * 2
            key = (tuple(towers), nl, n_img)
            buf = self._gnf_ws.get(key)
            if buf is None:
                buf = self._gnf_ws[key] = torch.empty((nt * spec.NUM_CONVS * per,), device=self.device, dtype=torch.float32)
            buf.zero_()
        for i in range(spec.NUM_CONVS):
            xs = [t[tw][l] for l in range(nl) for tw in towers]
            pcs = [cv["%s%s.%d" % (h, tw, 3 * i)].pc for l in range(nl) for tw in towers]
            gnb, wsl = None, {}
            if nf > 0:
                for k, tw in enumerate(towers):
                    wsl[tw] = buf[(k * spec.NUM_CONVS + i) * per:(k * spec.NUM_CONVS + i + 1) * per]
                parts = {tw: ops.gn_fwd_ws_parts(wsl[tw], nl, n_img, spec.GN_GROUPS) for tw in towers}
                gnb = {"wss": [parts[tw][l] if l < nf else None for l in range(nl) for tw in towers], "n": n_img, "groups": spec.GN_GROUPS}
            us = ops.conv2d_multi(xs, pcs, pad=1, gnb=gnb)
            for k, tw in enumerate(towers):
                (gw, _), (gbeta, _) = self.gn("%s%s.%d" % (h, tw, 3 * i + 1))
                u = us[k::nt]
                t2, ab = ops.groupnorm_relu_levels(u, gw, gbeta, spec.GN_GROUPS, spec.GN_EPS, ws=wsl.get(tw),
                                                   fused_mask=(1 << nf) - 1 if nf > 0 else 0)
                layers[tw].append((t[tw], u, ab))
                t[tw] = t2
        outs = {}
        prepack = not self.split_levels and hasattr(self, "pred_dgrad_weights")
        if "cls_tower" in towers:
            outs["cls_tower"] = ops.conv2d_grouped(t["cls_tower"], cv[h + "cls_ctr"].pc, pad=1)
            if prepack:
                self.pred_dgrad_weights(cv[h + "cls_ctr"])
        if "bbox_tower" in towers:
            outs["bbox_tower"] = ops.conv2d_grouped(t["bbox_tower"], cv[h + "bbox_pred"].pc, pad=1, act=ACT_EXP_SCALE,
                                                    act_scale_devs=[scales[l:l + 1] for l in lv])
            if prepack:
                self.pred_dgrad_weights(cv[h + "bbox_pred"])
        return outs, {tw: (layers[tw], t[tw]) for tw in towers}

    def loss_and_grads(self, head_out, gt_boxes, gt_count):
        """-> losses [4] (cls, reg, centerness, num_pos) on the device, per-level gradients w.r.t. the prediction convs."""
        h = "rpn.head."
        scales, gscales = self.extra[h + "scales"]
        n = head_out[0][0].shape[0]
        # one zeroed buffer for the loss sums and the Scale gradients' raw sums (one fill launch on the chain instead of two)
        zbuf = torch.zeros(16, device=self.device, dtype=torch.float32)
        sums = zbuf[:8]
        nl = len(head_out)
        ops.fcos_loss_levels(0, head_out, gt_boxes, gt_count, spec.FPN_STRIDES[:nl], SIZE_RANGES[:nl], spec.POS_RADIUS,
                             spec.LOSS_GAMMA, spec.LOSS_ALPHA, None, sums)
        gstride = self.convs[h + "bbox_pred"].pd.cin_k
        grads = []
        raw = zbuf[8:8 + max(nl, 5)]
        for lvl, (cc, rg) in enumerate(head_out):
            shape = cc.shape[:3] + (gstride,)
            # persistent gradient buffers: the kernel rewrites the real channels of EVERY location each step, the padding
            # channels (K padding of the data-gradient convs) are zeroed once here instead of by 10 fill launches per step
            key = (lvl, tuple(shape), self.dtype)
            if key not in self._pred_grad_bufs:
                self._pred_grad_bufs[key] = (torch.zeros(shape, device=self.device, dtype=self.dtype),
                                             torch.zeros(shape, device=self.device, dtype=self.dtype))
            grads.append(self._pred_grad_bufs[key])
        ops.fcos_loss_levels(1, head_out, gt_boxes, gt_count, spec.FPN_STRIDES[:nl], SIZE_RANGES[:nl], spec.POS_RADIUS,
                             spec.LOSS_GAMMA, spec.LOSS_ALPHA, [scales[l:l + 1] for l in range(nl)], sums,
                             [g[0] for g in grads], [g[1] for g in grads], [raw[l:l + 1] for l in range(nl)])
        losses = torch.empty(4, device=self.device, dtype=torch.float32)
        # the losses, and d loss / d scale_l = sum ds * x, x = log(reg) / scale_l: gscales += raw / scales in the same launch
        ops._lib.call("osd_fcos_loss_finalize_scales", ops._p(sums), ops._p(losses), n, ops._p(raw), ops._p(scales), ops._p(gscales), nl,
                      ops._stream())
        # {num_pos, sum_w, sum_focal, sum_w*(1-giou), sum_bce}: the un-normalised sums are additive over images (tests)
        self.last_loss_sums = sums
        return losses, grads
