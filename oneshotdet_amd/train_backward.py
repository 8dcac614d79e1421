"""Hand-scheduled backward pass (TrainEngine mixin): data gradients on the forward implicit-GEMM kernels with flipped weights,
weight gradients queued per stage / tower and launched on side streams, GroupNorm backward, gradient buckets announced to the
exchange as soon as their last writer is enqueued.  The reference gets all of this from autograd (engine/trainer.py:79-93)."""

import torch

from . import streams

from . import ops, spec


class BackwardPass(object):
    def _wstream_of(self, which):
        """Weight-gradient stream 0 (target backbone, cls tower) or 1 (query backbone, bbox tower); None = inline."""
        if self.wstream is None:
            return None
        return self.wstream2 if (which == 1 and self.wstream2 is not None) else self.wstream

    def _on_wstream(self, fn, tensors, which=0):
        ws = self._wstream_of(which)
        if ws is None:
            return fn()
        ev = torch.cuda.Event()
        ev.record()
        ws.wait_event(ev)
        with streams.on(ws):
            fn()
        self._keep.append(tensors)        # keep the operands alive until the side stream has been joined

    def _bucket_ready(self, name, which=0, extra=()):
        """Everything that writes gradient bucket `name` has been enqueued (weight gradients on side stream `which`; for
        the head also the GroupNorm / Scale gradients on the compute stream): start its all-reduce behind those streams
        and, inside train_step, its SGD update + repack behind that.  The update also waits for the current compute
        stream: the bucket's data-gradient convs (enqueued before this point) read the packed weights it rewrites."""
        if not self._overlap or name is None:
            return
        cur = streams.current()
        ws = self._wstream_of(which)
        producers = [cur if ws is None else ws] + list(extra)
        if self._fuse_update and cur not in producers:
            producers.append(cur)
        if self.exchange.active:
            self.exchange.ready(name, producers)
        if not self._fuse_update:
            return
        ust = self.exchange.comm if self.exchange.active else self.ustream
        if not self.exchange.active:
            for st in producers:
                ev = torch.cuda.Event()
                ev.record(st)
                ust.wait_event(ev)
        with streams.on(ust):
            self._update_bucket(name)

    def _flush_wgrads(self, j, which):
        """Launch the queued weight gradients of backbone j's stage as ONE mixed-geometry launch (<= 24 convs each) on side
        stream `which`:
        all output tiles share the workgroup budget in proportion to their work, so every conv runs with few pixel splits
        — long inner loops, little atomic traffic — and the query branch's latency-sized launches disappear into it."""
        q, self._wqs[j] = self._wqs[j], []
        for i in range(0, len(q), 24):
            part = q[i:i + 24]
            if len(part) == 1:
                c, x, dy, stride, pad = part[0]
                self._on_wstream(lambda c=c, x=x, dy=dy, stride=stride, pad=pad: ops.conv2d_wgrad(
                    x, dy, c.gw, c.r, c.s, stride, pad, c.cout, scale=c.bn_scale, db=c.gb if c.has_bias else None), (x, dy),
                    which)
            else:
                items = [(x, dy, c.gw, c.bn_scale, c.gb if c.has_bias else None, c.r, c.s, stride, pad, c.cout)
                         for c, x, dy, stride, pad in part]
                self._on_wstream(lambda items=items: ops.conv2d_wgrad_mixed(items), items, which)

    def _wgrad_grouped(self, c, pairs, which=0, g=None):
        self._on_wstream(lambda: ops.conv2d_wgrad_grouped(pairs, c.gw, c.r, c.s, 1, c.r // 2, c.cout, scale=c.bn_scale,
                                                          db=c.gb if c.has_bias else None, g=g), (pairs, g), which)

    def _pred_backward(self, c, t_last, dpred, which):
        """Weight AND data gradient of a prediction conv (cls_logits + centerness fused / bbox_pred, fcos.py:50-61,91-97) over the FPN
        levels from ONE gathered matrix G [pixels][64] (the nine shifted dy vectors of every pixel side by side): the weight
        gradient as the 1x1 problem (t_last, G) on the side stream, the data gradient as a 1x1 conv with K = 64 over G — one launch
        that writes all levels (round 5; until then a 3x3 conv over dy with its 2 / 4 input channels padded to 64 per tap:
        118 - 132 us on the critical chain for a 70 MB write).  -> the per-level d(t_last), views of one buffer."""
        nl = len(dpred)
        g = ops.pred_dy_gather(dpred, c.cout, c.cin)
        self._wgrad_grouped(c, [(t_last[l], dpred[l]) for l in range(nl)], which, g=g)
        ent = self.pred_dgrad_weights(c, fresh_ok=True)
        dx = ops.conv2d(g.view(1, 1, g.shape[0], ops.PRED_G), ent)
        out, q0 = [], 0
        for t in dpred:
            n, hh, ww = t.shape[0], t.shape[1], t.shape[2]
            out.append(dx.view(-1, c.cin)[q0:q0 + n * hh * ww].view(n, hh, ww, c.cin))
            q0 += n * hh * ww
        return out

    def pred_dgrad_weights(self, c, fresh_ok=False):
        """The prediction conv's weights in the gathered data-gradient form (osd_pred_dgrad_pack: 256 x 64 values, from the master that
        changes every step).  The FORWARD pass calls this right behind the prediction conv, on that conv's stream (round 6): between the
        loss and the backward pass the chain is a string of small launches with nothing running beside them, and this one needs nothing
        from the loss.  fresh_ok: the backward pass's call — packs only if this step's forward did not."""
        key = ("pred_dgrad", c.name)
        ent = self._pred_dgrad.get(key)
        if ent is None:
            wd = torch.empty((c.cin, 1, 1, ops.PRED_G), device=self.device, dtype=self.dtype)
            ent = self._pred_dgrad[key] = ops.PackedConv(wd, torch.zeros(ops._round_up(c.cin, 16), device=self.device, dtype=torch.float32),
                                                         c.cin, c.cin, c.cin, ops.PRED_G, 1, 1, cin_real=9 * c.cout)
        fresh = getattr(self, "_pred_dgrad_fresh", None)
        if fresh is None:
            fresh = self._pred_dgrad_fresh = set()
        if fresh_ok and c.name in fresh:
            fresh.discard(c.name)
            return ent
        ops.pred_dgrad_pack(c.w, c.cout, c.cin, self.dtype, out=ent.w)
        if not fresh_ok:
            fresh.add(c.name)
        return ent

    def _dgrad_levels(self, c, dys):
        """Data gradient of a conv shared by the FPN levels: one grouped launch (forward kernel, flipped weights)."""
        return ops.conv2d_grouped(dys, c.pd, pad=c.r - 1 - (c.r // 2))

    def head_backward(self, feats, ctxs, pred_grads):
        """Layer by layer (last first): GroupNorm+ReLU backward of each tower (two launches for its five levels), then the
        data gradient of BOTH towers' conv over all levels as ONE launch; the weight gradients of a tower's four convs x
        five levels go out as one launch on that tower's side stream once its chain is done."""
        nl = len(feats)
        if self.towers_merged:
            d_t = self._towers_backward(ctxs, pred_grads, self.TOWERS, nl)
        else:
            main = streams.current()
            side = self.s1 if self.s1 is not None else main
            side.wait_stream(main)
            with streams.on(side):
                d_t = self._towers_backward(ctxs, pred_grads, ("bbox_tower",), nl)
            if self.fuse_head_sum:
                # d combined = d(cls tower input) + d(bbox tower input): the cls tower's LAST data-gradient conv takes the bbox
                # tower's as its residual operand (the epilogue's RES_SAME add), so the sum costs neither a launch per level nor
                # a write + re-read of the five level maps (round 3: 5 x add_mask, 0.67 ms of kernel time inside the step)
                def bbox_grads():
                    main.wait_stream(side)
                    self._keep.append(d_t["bbox_tower"])
                    return d_t["bbox_tower"]
                return self._towers_backward(ctxs, pred_grads, ("cls_tower",), nl, last_addends=bbox_grads)["cls_tower"]
            d_t.update(self._towers_backward(ctxs, pred_grads, ("cls_tower",), nl))
            main.wait_stream(side)
        return [ops.add_mask(d_t["cls_tower"][l], d_t["bbox_tower"][l]) for l in range(nl)]

    def _towers_backward(self, ctxs, pred_grads, towers, nl, last_addends=None):
        cv = self.convs
        h = "rpn.head."
        nt = len(towers)
        d_t, items = {}, {tw: [] for tw in towers}
        for tw in towers:
            k = self.TOWERS.index(tw)
            layers, t_last = ctxs[tw]
            pc = cv[h + ("cls_ctr" if tw == "cls_tower" else "bbox_pred")]
            dpred = [pred_grads[l][k] for l in range(nl)]
            if self.pred_dgrad_gemm and ops.pred_gemm_ok(pc.cout, pc.r, pc.s, 1, pc.r // 2, pc.cin, nl):
                d_t[tw] = self._pred_backward(pc, t_last, dpred, k)
            else:
                self._wgrad_grouped(pc, [(t_last[l], dpred[l]) for l in range(nl)], k)
                d_t[tw] = self._dgrad_levels(pc, dpred)
        # GroupNorm-backward statistics of layer i - 1 are gathered by the epilogue of the data-gradient conv of layer i (the
        # conv that writes the gradient w.r.t. that GroupNorm's output) where the kernel can (ops.gn_bwd_fusable: the large
        # levels); the GroupNorm backward of those levels then is one pass over (u, dt) instead of two.  Not in ordered mode:
        # the sums are added atomically
        fuse = self.fuse_gn_bwd and not any(isinstance(ctxs[tw][0][0][2], list) for tw in towers)
        fused = {tw: 0 for tw in towers}        # levels of d_t[tw] whose sums are already in that layer's workspace
        if fuse:
            u0 = ctxs[towers[0]][0][0][1]
            n_img, c_gn = u0[0].shape[0], u0[0].shape[-1]
            numel = ops.gn_bwd_ws_numel(nl, n_img, c_gn, spec.GN_GROUPS)
            key = (tuple(towers), nl, n_img, c_gn)
            if self._gnb_ws.get("key") != key:
                self._gnb_ws = {"key": key, "buf": torch.empty((len(towers) * spec.NUM_CONVS * numel,), device=self.device, dtype=torch.float32)}
            buf = self._gnb_ws["buf"]
            buf.zero_()
            gws = {(tw, i): buf[(k * spec.NUM_CONVS + i) * numel:(k * spec.NUM_CONVS + i + 1) * numel]
                   for k, tw in enumerate(towers) for i in range(spec.NUM_CONVS)}
        for i in range(spec.NUM_CONVS - 1, -1, -1):
            dus = {}
            for tw in towers:
                (gw, ggw), (gbeta, ggb) = self.gn("%s%s.%d" % (h, tw, 3 * i + 1))
                c = cv["%s%s.%d" % (h, tw, 3 * i)]
                t_in, u, ab = ctxs[tw][0][i]
                # the tower conv's bias gradient (sum of du) comes out of the GroupNorm backward's own sums; the weight-gradient
                # launch then runs without its d-bias column sums (ops.groupnorm_relu_bwd_levels(conv_db=...))
                gn_db = c.gb if (self.gn_conv_db and c.has_bias and not fuse) else None
                if isinstance(ab, list):        # forward ran one chain per level group: one saved-statistics block each
                    dus[tw], lo = [], 0
                    for ab_g in ab:
                        k = ab_g.shape[0]
                        dus[tw] += ops.groupnorm_relu_bwd_levels(u[lo:lo + k], d_t[tw][lo:lo + k], ab_g, gw, gbeta, ggw, ggb, spec.GN_GROUPS,
                                                                 conv_db=gn_db)
                        lo += k
                else:
                    dus[tw] = ops.groupnorm_relu_bwd_levels(u, d_t[tw], ab, gw, gbeta, ggw, ggb, spec.GN_GROUPS,
                                                            ws=gws[(tw, i)] if fuse else None, fused_mask=fused[tw], conv_db=gn_db)
                items[tw] += [(t_in[l], dus[tw][l], c.gw, c.bn_scale, c.gb if (c.has_bias and gn_db is None) else None) for l in range(nl)]
            dys = [dus[tw][l] for l in range(nl) for tw in towers]
            c0 = cv["%s%s.%d" % (h, towers[0], 3 * i)]
            pds = [cv["%s%s.%d" % (h, tw, 3 * i)].pd for l in range(nl) for tw in towers]
            pad = c0.r - 1 - (c0.r // 2)
            gnb = None
            if fuse and i > 0:
                nf = 0          # leading levels the kernel can gather the sums of
                while nf < nl and all(ops.gn_bwd_fusable(dus[tw][nf], cv["%s%s.%d" % (h, tw, 3 * i)].pd, 1, pad) for tw in towers):
                    nf += 1
                if nf > 0:
                    gnb = {"us": [], "abs": [], "gammas": [], "wss": [], "pws": [], "n": n_img, "groups": spec.GN_GROUPS}
                    for l in range(nl):
                        for tw in towers:
                            _, u_prev, ab_prev = ctxs[tw][0][i - 1]
                            (gw_prev, _), _ = self.gn("%s%s.%d" % (h, tw, 3 * (i - 1) + 1))
                            ws_l, pw_l = ops.gn_bwd_ws_parts(gws[(tw, i - 1)], nl, n_img, c_gn, spec.GN_GROUPS)[l]
                            on = l < nf
                            gnb["us"].append(u_prev[l] if on else None)
                            gnb["abs"].append(ab_prev[l] if on else None)
                            gnb["gammas"].append(gw_prev if on else None)
                            gnb["wss"].append(ws_l if on else None)
                            gnb["pws"].append(pw_l if on else None)
                fused = {tw: (1 << nf) - 1 for tw in towers}
            addends = last_addends() if (i == 0 and last_addends is not None) else None      # (one tower per call: level order)
            out = ops.conv2d_multi(dys, pds, pad=pad, gnb=gnb, residuals=addends)
            for k, tw in enumerate(towers):
                d_t[tw] = out[k::nt]
        for tw in towers:
            c0 = cv["%s%s.0" % (h, tw)]
            job = (lambda it=items[tw], c0=c0: ops.conv2d_wgrad_multi(it, c0.r, c0.s, 1, c0.r // 2, c0.cout), items[tw], self.TOWERS.index(tw))
            if self.tower_wgrad_at:           # experiment: hold the towers' weight gradients until the backbone backward reaches a stage
                self._held_wgrads.append(job)
            else:
                self._on_wstream(*job)
        return d_t

    def _release_held_wgrads(self, point):
        """OSD_TOWER_WGRAD_AT=<point> (experiment): the towers' 252-workgroup team-mode weight gradients take every CU for
        ~0.6 ms each; launched where the head backward ends they stall the full-chip kernels behind them (the last tower data
        gradient, the correlation backward, FPN P3); launched at `point` of the TARGET backbone's backward they run beside
        layer4 / layer3 kernels that fill 100 - 200 of the 256 CUs."""
        if self.tower_wgrad_at != point or not self._held_wgrads:
            return
        held, self._held_wgrads = self._held_wgrads, []
        cur = streams.current()
        for job in held:
            self._on_wstream(*job)
        self._bucket_ready("head", 0, [st for st in (cur, self.wstream, self.wstream2) if st is not None])

    def backbones_backward(self, ctxs, dPs, which0=0):
        """Backward of both backbones in lockstep (the mirror of backbones_forward): every data-gradient conv is ONE launch
        over (target, query); the weight gradients are queued per backbone and go out per stage as mixed-geometry launches
        on that backbone's side stream; a stage's gradient bucket is announced as soon as its last writer is enqueued."""
        cv = self.convs
        bbs = [c["bb"] for c in ctxs]
        nb = len(ctxs)
        self._wqs = [[] for _ in ctxs]

        def col(key):
            return [c[key] for c in ctxs]

        def W(name, xs, dys, stride=1, pad=0):
            for j in range(nb):
                self._wqs[j].append((cv[bbs[j] + name], xs[j], dys[j], stride, pad))

        def D(name, dys, residuals=None, masks=None):
            """Data gradient of a stride-1 conv: the forward kernel on dy with flipped/transposed weights."""
            c = cv[bbs[0] + name]
            return ops.conv2d_multi(dys, [cv[bb + name].pd for bb in bbs], pad=c.r - 1 - (c.r // 2), residuals=residuals,
                                    masks=masks)
        f = "fpn."
        dp3, dp4, dp5, dp6, dp7 = ([dP[l] for dP in dPs] for l in range(5))
        # P7 = conv(relu(P6)), P6 = conv(P5), both 3x3 stride 2 (fpn.py:95-99); 3x3 stride-2 data gradient = zero-insert
        # dY to the input grid, then the stride-1 flipped-weight conv
        W(f + "top_blocks.p7", col("p6r"), dp7, 2, 1)
        t = D(f + "top_blocks.p7", [ops.scatter2x(d, p6.shape[1:3]) for d, p6 in zip(dp7, col("p6"))], masks=col("p6"))
        d_p6 = [ops.add_mask(a, b) for a, b in zip(t, dp6)]
        W(f + "top_blocks.p6", col("p5"), d_p6, 2, 1)
        d_p5 = D(f + "top_blocks.p6", [ops.scatter2x(d, p5.shape[1:3]) for d, p5 in zip(d_p6, col("p5"))], residuals=dp5)
        W(f + "fpn_layer4", col("inner4"), d_p5, 1, 1)
        W(f + "fpn_layer3", col("inner3"), dp4, 1, 1)
        W(f + "fpn_layer2", col("inner2"), dp3, 1, 1)
        if self.fpn_out_grouped:        # the data gradients of the P3 and P4 output convs as one launch (see the forward pass)
            c_ = cv[bbs[0] + f + "fpn_layer2"]
            d23 = ops.conv2d_multi(dp3 + dp4, [cv[bb + f + "fpn_layer2"].pd for bb in bbs] + [cv[bb + f + "fpn_layer3"].pd for bb in bbs],
                                   pad=c_.r - 1 - (c_.r // 2))
            d_inner2, d_inner3 = d23[:nb], d23[nb:]
        else:
            d_inner2 = D(f + "fpn_layer2", dp3)
            d_inner3 = D(f + "fpn_layer3", dp4)
        d_inner3 = [ops.upsample2x_bwd(a, b) for a, b in zip(d_inner2, d_inner3)]
        d_inner4 = D(f + "fpn_layer4", d_p5)
        d_inner4 = [ops.upsample2x_bwd(a, b) for a, b in zip(d_inner3, d_inner4)]
        W(f + "fpn_inner4", col("c5"), d_inner4)
        W(f + "fpn_inner3", col("c4"), d_inner3)
        W(f + "fpn_inner2", col("c3"), d_inner2)
        # gradients w.r.t. C5 / C4 / C3 from the laterals; C5's is complete, so its ReLU mask is applied here
        g = D(f + "fpn_inner4", d_inner4, masks=col("c5"))
        lat4, lat3 = D(f + "fpn_inner3", d_inner3), D(f + "fpn_inner2", d_inner2)
        lateral = {}
        for j in range(nb):
            lateral[id(ctxs[j]["c4"])] = lat4[j]
            lateral[id(ctxs[j]["c3"])] = lat3[j]
        if which0 == 0:
            self._release_held_wgrads("fpn")
        # body, last block first.  `g` = gradient w.r.t. the block output, already masked by its ReLU.
        for bi in range(len(ctxs[0]["blocks"]) - 1, -1, -1):
            blks = [c["blocks"][bi] for c in ctxs]
            p, s, has_ds = blks[0]["p"], blks[0]["s"], blks[0]["ds"]
            bx, bo1, bo2 = ([b[k] for b in blks] for k in ("x", "o1", "o2"))
            W(p + "conv3", bo2, g)
            d_o2 = D(p + "conv3", g, masks=bo2)
            W(p + "conv2", bo1, d_o2, 1, 1)
            d_o1 = D(p + "conv2", d_o2, masks=bo1)
            W(p + "conv1", bx, d_o1, s, 0)
            if has_ds:
                W(p + "downsample.0", bx, g, s, 0)
            stage = p[len("body."):].split(".", 1)[0]
            sname = ("layer4+fpn" if stage == "layer4" else stage) if p.endswith(".0.") else None

            def stage_done():
                for j in range(nb):
                    self._flush_wgrads(j, which0 + j)
                    self._bucket_ready(bbs[j].rstrip(".") + "." + sname, which0 + j)
                if which0 == 0:
                    self._release_held_wgrads(stage)
            if blks[0]["first"]:
                stage_done()
                break                                   # input of layer2 = frozen layer1 output: no data gradient
            extra = [lateral.get(id(x)) for x in bx]    # block input is C3/C4: add the FPN lateral's gradient
            has_extra = extra[0] is not None
            if s == 1:
                if has_ds:
                    a = D(p + "downsample.0", g, residuals=extra if has_extra else None)
                else:
                    a = [ops.add_mask(gg, e) for gg, e in zip(g, extra)] if has_extra else g
                g = D(p + "conv1", d_o1, residuals=a, masks=bx)
            else:                                        # 1x1 stride 2: small-grid GEMM, then zero-insert
                a = D(p + "downsample.0", g)
                bsm = D(p + "conv1", d_o1, residuals=a)
                g = [ops.scatter2x(b_, x.shape[1:3], mask=x, addend=e) for b_, x, e in zip(bsm, bx, extra)]
            if sname is not None:       # first block of its stage done (its data-gradient convs included): the stage's
                stage_done()            # weight gradients go out, then its gradients are final and nothing enqueued
        for j in range(nb):             # later reads its packed weights
            self._flush_wgrads(j, which0 + j)
        self._wqs = None
        if which0 == 0:
            self._release_held_wgrads(self.tower_wgrad_at)      # whatever was not released on the way
        return None
