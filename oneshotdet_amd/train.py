"""Training step of the siamese-FCOS hot path on MI355X: forward with saved activations, FCOS loss, hand-scheduled
backward (data gradients on the forward implicit-GEMM kernel with flipped weights, weight gradients on the MFMA
wgrad kernel, GroupNorm / ROIAlign / correlation backward), gradient all-reduce over RCCL and SGD.

Reference call stack: engine/trainer.py:62-96 -> GeneralizedRCNN.forward (train) ->
FCOSModule._forward_train (fcos.py:178-200) -> FCOSLossComputation.__call__ (fcos/loss.py:213-276); autograd does the
backward there.  Stem and layer1 are frozen (resnet.py:127-136, FREEZE_CONV_BODY_AT=2) and FrozenBN has no parameters,
so gradients stop at the input of layer2.

Master weights are fp32 and live, like their gradients, in ONE flat buffer each, conv weights in the kernels' own
[Cout][R][S][Cin] order: the optimiser (elementwise), the weight-gradient kernel (atomics into that layout) and the
bf16/fp32 packers all share it, and the gradient all-reduce is a handful of large contiguous RCCL calls.
`state_dict()` converts back to the reference's OIHW names/shapes.
"""
import math
import os

import torch

from . import ops, spec
from .ops import ACT_EXP_SCALE, ACT_NONE, ACT_RELU, RES_NONE, RES_SAME, RES_UP2X, PackedConv

SIZE_RANGES = ((-1.0, 64.0), (64.0, 128.0), (128.0, 256.0), (256.0, 512.0), (512.0, float(spec.INF)))


def _payload(t):
    """The device tensor inside a batch as the pipeline hands it over (plain tensor, layers.ImageList, ops.PackedImages)."""
    from .layers import ImageList
    if isinstance(t, ops.PackedImages):
        return t.tensor
    if isinstance(t, ImageList):
        return t.tensors
    return t


def _static_clone(t):
    from .layers import ImageList
    if isinstance(t, ops.PackedImages):
        return ops.PackedImages(t.tensor.clone(), t.hw, t.image_sizes)
    if isinstance(t, ImageList):
        return ImageList(t.tensors.clone(), list(t.image_sizes))
    return t.clone()


def _static_copy(dst, src):
    """New contents for a captured graph's static input.  The geometry (batch shape, every image's true size) is part of
    the capture: a batch that differs in it needs a new capture()."""
    for attr in ("image_sizes", "hw"):
        a, b = getattr(dst, attr, None), getattr(src, attr, None)
        if a is not None and b is not None and [tuple(v) for v in ([a] if attr == "hw" else a)] != \
                [tuple(v) for v in ([b] if attr == "hw" else b)]:
            raise ValueError("replay_step: the batch's %s differ from the captured ones; call capture() again" % attr)
    d, s_ = _payload(dst), _payload(src)
    if d.shape != s_.shape or d.dtype != s_.dtype:
        raise ValueError("replay_step: input %s %s does not match the captured %s %s" % (tuple(s_.shape), s_.dtype, tuple(d.shape), d.dtype))
    d.copy_(s_, non_blocking=True)


class TConv(object):
    """One convolution of the training graph."""

    def __init__(self, name, cout, cin, r, s, trainable, has_bias):
        self.name, self.cout, self.cin, self.r, self.s = name, cout, cin, r, s
        self.trainable, self.has_bias = trainable, has_bias
        self.w = self.gw = self.b = self.gb = None       # views into the flat master / gradient buffers
        self.bn_scale = self.bn_shift = None             # folded FrozenBN (fp32 device tensors)
        self.pc = None                                   # forward PackedConv
        self.pd = None                                   # data-gradient PackedConv (flipped, transposed weights)
        self.need_dgrad = True


class TrainEngine(object):
    def __init__(self, state_dict, dtype=torch.bfloat16, device="cuda", lr=0.0005, momentum=0.9, weight_decay=0.0001,
                 process_group=None, wgrad_side_stream=True, optimizer="fused", second_stage=False, ordered_wgrad=None,
                 exchange_single_rank=False, grad_wire_dtype=None):
        if not torch.cuda.is_available():
            raise ops._lib.OsdError("TrainEngine needs an MI355X: no GPU visible and there is no CPU fallback")
        ops._lib.load()
        self.device, self.dtype = torch.device(device), dtype
        self.pg = process_group
        # weight gradients feed nothing but the optimiser: they run on a side stream, beside the data-gradient chain
        self.wstream = torch.cuda.Stream(device=self.device) if wgrad_side_stream else None
        # second stream: the pooled-query gradient chain (a dozen latency-sized kernels) beside the correlation backward.
        # The query backbone and the bbox tower need no stream of their own: they ride in the launches of the target
        # backbone / cls tower (osd_conv2d_fwd_multi)
        self.s1 = torch.cuda.Stream(device=self.device) if wgrad_side_stream else None
        self.wstream2 = torch.cuda.Stream(device=self.device) if wgrad_side_stream else None   # query backbone / bbox tower
        # training proposals feed only the second stage: a chain of small kernels on a stream of their own
        self.pstream = torch.cuda.Stream(device=self.device) if wgrad_side_stream else None
        self._keep = []
        self.second_stage = bool(second_stage)
        self.box_keys, self.box_losses = None, None       # sampler keys for the next step (None: torch.rand), last losses
        from .model import ProposalDepth
        self._prop_depth = None if os.environ.get("OSD_NO_PROP_HINT") else ProposalDepth()
        # A/B switches (measured on one box, tools/ab_bench.sh): both backbones per launch / both towers per launch
        self.lockstep = os.environ.get("OSD_LOCKSTEP", "0") != "0"
        self.corr_levels = os.environ.get("OSD_CORR_LEVELS", "1") != "0"
        self.skip_unused_c2 = os.environ.get("OSD_FULL_C2", "0") == "0"
        self.towers_merged = os.environ.get("OSD_TOWERS_MERGED", "0") != "0"
        # A/B: P3+P4 and P5-P7 of a tower as separate forward chains on the (then idle) weight-gradient streams.  Measured SLOWER
        # (12.05 vs 11.90 ms per step, same box): like lockstep / merged towers, more concurrency buys nothing here
        self.split_levels = os.environ.get("OSD_SPLIT_LEVELS", "0") != "0"
        # the sum of the two towers' input gradients inside the cls tower's last data-gradient conv (A/B: OSD_NO_HEAD_SUM_FUSION=1)
        self.fuse_head_sum = os.environ.get("OSD_NO_HEAD_SUM_FUSION", "0") == "0"
        sd = {k: torch.as_tensor(v).to(self.device, torch.float32) for k, v in state_dict.items()}
        self._frozen_sd = sd
        self.convs = {}          # name -> TConv
        self.extra = {}          # name -> (param view, grad view) for GN affine and Scale parameters
        self._plan = []          # (name, shape) of every trainable tensor, in registration order
        self._build(sd)
        self._allocate(sd)
        from . import model as _model
        self._fused_l1 = {}      # backbone prefix -> conv3 + downsample of the frozen layer1.0 as one packed 1x1 conv
        if _model.FUSE_DOWNSAMPLE and dtype == torch.bfloat16:       # see model.BackboneWeights: bf16 engines only
            for bb in self.BBS:
                if (bb + "body.layer1.0.downsample.0.weight") in sd:
                    self._fused_l1[bb] = _model.pack_conv3_downsample(sd, bb + "body.layer1.0.", dtype)
        # gradient exchange overlapped with backward (no-op with one rank): buckets in the order they become final; the
        # optimiser update + weight repack of a bucket follow its exchange on the same side stream (train_step)
        from .dist_utils import GradExchange, bucket_ranges
        # exchange_single_rank: run the collectives with ONE rank too (the step as it runs under RCCL, measurable on a one-GPU
        # box); grad_wire_dtype=torch.bfloat16: half-width buckets on the wire (dist_utils.GradExchange)
        self.ustream = torch.cuda.Stream(device=self.device)
        self.exchange = GradExchange(self.flat_g, bucket_ranges(self._plan, self.flat_g.numel()), self.pg,
                                     single_rank_too=exchange_single_rank, wire_dtype=grad_wire_dtype, comm_stream=self.ustream)
        self._overlap, self._fuse_update, self._updated = True, False, set()
        # consume_grads: the fused update overwrites the gradients it has read with zeros, so train_step() leaves flat_g clean
        # for the next step instead of the next forward pass memset-ing 236 MB on its critical path.  named_grads() after a
        # train_step() then reads zeros: tests that inspect gradients use forward_backward(), or switch this off
        self.consume_grads = os.environ.get("OSD_NO_CONSUME_GRADS", "0") == "0"
        self._grads_clean, self._zeroed = False, set()
        self._wqs = None
        self._pred_grad_bufs = {}
        # ordered weight gradients (bit-reproducible dW of every conv_wgrad launch; measured 6 % slower on the tower launch,
        # 20-55 % on the short ones: off by default, OSD_WGRAD_ORDERED=1 or ordered_wgrad=True): 1 GiB of scratch per stream
        if ordered_wgrad is None:
            ordered_wgrad = os.environ.get("OSD_WGRAD_ORDERED", "0") != "0"
        self.ordered_wgrad = bool(ordered_wgrad)
        # GroupNorm-backward statistics gathered by the data-gradient convs' epilogues (atomics: not in ordered mode).  Built,
        # parity-tested and OFF: the epilogue's extra read of u costs the conv more (+65 us per launch over P3 + P4 of both
        # towers, +130 with the arithmetic and the atomics) than the statistics pass it replaces (2 x 26 us, at the HBM
        # roofline); 11.70 vs 11.34 ms per step in same-box A/B.  OSD_GN_FUSION=1 turns it on
        self.fuse_gn_bwd = (not self.ordered_wgrad and self.dtype == torch.bfloat16 and os.environ.get("OSD_GN_FUSION", "0") != "0")
        self._gnb_ws = {}
        # the forward statistics (sum, sum of squares of the tower conv's outputs) in the forward conv's epilogue: no extra operand
        # there, two atomics per wave and group — and still a loss: the conv over P3 + P4 of both towers 243 -> 275 us to save
        # 2 x 13 us of statistics pass, 11.85 vs 11.55 ms per step in same-box A/B.  OSD_GN_FWD_FUSION=1 turns it on
        self.fuse_gn_fwd = (not self.ordered_wgrad and self.dtype == torch.bfloat16 and os.environ.get("OSD_GN_FWD_FUSION", "0") != "0")
        self._gnf_ws = {}
        # torch hands out stream handles from a pool, so a handle may carry an earlier engine's registration: set the mode of
        # this engine's weight-gradient streams explicitly either way (close() / __del__ release the scratch buffers)
        # (the second stage's weight gradients run on the proposal stream: box_head_forward_backward)
        cands = (self.wstream, self.wstream2) + ((self.pstream,) if self.second_stage else ())
        self._wgrad_streams = list({id(x): x for x in cands if x is not None}.values()) or [torch.cuda.current_stream()]
        if self.ordered_wgrad and self.second_stage:
            import warnings
            warnings.warn("ordered_wgrad makes every conv_wgrad launch bit-reproducible, the second stage's included; its "
                          "ROI-pool backward (osd_roi_pool_levels_bwd) still scatters with fp32 atomics, so the gradients it "
                          "feeds into both backbones are reproducible only up to the order of those adds")
        for st in self._wgrad_streams:
            ops.wgrad_set_workspace(st, (1 << 30) if self.ordered_wgrad else 0)
        self.defer_join = False       # opt-in: train_step leaves its tail on the side streams (see train_step / join)
        self._deferred, self._defer_now, self._joined_refs = None, False, None
        self.repack()
        self.zero_bias = torch.zeros(4096, device=self.device, dtype=torch.float32)
        # SGD with the reference's parameter groups (solver/build.py:8-26): a parameter whose reference KEY contains "bias"
        # gets lr x BIAS_LR_FACTOR (2) and WEIGHT_DECAY_BIAS (0); every other one — conv / GroupNorm weights AND the Scale
        # scalars `rpn.head.scales.N.scale` — gets BASE_LR and WEIGHT_DECAY
        weights = [c.w for c in self.convs.values() if c.trainable] + \
                  [p for n, (p, _) in self.extra.items() if "bias" not in n]
        biases = [c.b for c in self.convs.values() if c.trainable and c.has_bias] + \
                 [p for n, (p, _) in self.extra.items() if "bias" in n]
        self.lr, self.momentum, self.weight_decay = lr, momentum, weight_decay
        self.opt = None
        if optimizer == "torch":     # the reference's optimiser object, kept for A/B tests of the fused kernel
            self.opt = torch.optim.SGD([{"params": weights, "lr": lr, "weight_decay": weight_decay},
                                        {"params": biases, "lr": 2 * lr, "weight_decay": 0.0}], lr=lr, momentum=momentum)
        else:
            self._build_sgd_table(weights, biases)

    # ------------------------------------------------------------------------------------------------ construction
    def _add_conv(self, name, sd, bn=None, bias=None, trainable=True):
        w = sd[name + ".weight"]
        cout, cin, r, s = w.shape
        c = TConv(name, cout, cin, r, s, trainable, bias is not None)
        if bn is not None:
            g, b, mean, var = (sd[bn + k] for k in (".weight", ".bias", ".running_mean", ".running_var"))
            c.bn_scale = (g * var.rsqrt()).contiguous()          # layers/batch_norm.py:20 (no eps)
            c.bn_shift = (b - mean * c.bn_scale).contiguous()
        self.convs[name] = c
        if trainable:
            self._plan.append((name + ".weight", (cout, r, s, cin)))
            if bias is not None:
                self._plan.append((name + ".bias", (cout,)))
        return c

    def _build(self, sd):
        for bb in ("backbone.", "supp_backbone."):
            b = bb + "body."
            self._add_conv(b + "stem.conv1", sd, bn=b + "stem.bn1", trainable=False)
            for si, nblocks in enumerate(spec.STAGE_BLOCKS):
                for bi in range(nblocks):
                    p = "%slayer%d.%d." % (b, si + 1, bi)
                    tr = si >= 1
                    if (p + "downsample.0.weight") in sd:
                        self._add_conv(p + "downsample.0", sd, bn=p + "downsample.1", trainable=tr)
                    for i in (1, 2, 3):
                        self._add_conv("%sconv%d" % (p, i), sd, bn="%sbn%d" % (p, i), trainable=tr)
            f = bb + "fpn."
            for nm in ("fpn_inner2", "fpn_inner3", "fpn_inner4", "fpn_layer2", "fpn_layer3", "fpn_layer4", "top_blocks.p6",
                       "top_blocks.p7"):
                self._add_conv(f + nm, sd, bias=True)
        h = "rpn.head."
        for tower in ("cls_tower", "bbox_tower"):
            for i in range(spec.NUM_CONVS):
                self._add_conv("%s%s.%d" % (h, tower, 3 * i), sd, bias=True)
                self._plan.append(("%s%s.%d.weight" % (h, tower, 3 * i + 1), (spec.FPN_OUT,)))
                self._plan.append(("%s%s.%d.bias" % (h, tower, 3 * i + 1), (spec.FPN_OUT,)))
        # cls_logits + centerness fused into one 2-output conv (both read the cls tower, fcos.py:91-92)
        c = TConv(h + "cls_ctr", 2, spec.FPN_OUT, 3, 3, True, True)
        self.convs[h + "cls_ctr"] = c
        self._plan.append((h + "cls_ctr.weight", (2, 3, 3, spec.FPN_OUT)))
        self._plan.append((h + "cls_ctr.bias", (2,)))
        self._add_conv(h + "bbox_pred", sd, bias=True)
        self._plan.append((h + "scales", (5,)))
        if self.second_stage:
            self._build_box_head(sd)

    def _build_box_head(self, sd):
        """roi_heads.box.* (modeling/roi_heads/box_head/box_head.py:36-79) as this build lays it out: the first 1x1 conv split
        at the concatenation into its ROI half (no bias) and its query half (+ bias), fc6 as a 1x1 conv over the (h, w, c)
        flattening of the NHWC ROI maps, cls_score + bbox_pred as one 10-row conv.  Appended to the plan after the FCOS head:
        one more gradient bucket ('box_head')."""
        b = "roi_heads.box."
        missing = [k for k in spec.box_head_shapes() if k not in sd]
        if missing:
            raise KeyError("second_stage=True needs the roi_heads.box.* entries, e.g. %s" % missing[:2])
        c, mid, p = spec.FPN_OUT, spec.FPN_OUT // 2, spec.BOX_POOL

        def conv(name, cout, cin, r, s, has_bias):
            t = TConv(name, cout, cin, r, s, True, has_bias)
            self.convs[name] = t
            self._plan.append((name + ".weight", (cout, r, s, cin)))
            if has_bias:
                self._plan.append((name + ".bias", (cout,)))
        conv(b + "compress_dim_conv.0x", 2 * c, c, 1, 1, False)
        conv(b + "compress_dim_conv.0q", 2 * c, c, 1, 1, True)
        for gn_name in ("compress_dim_conv.1",):
            self._plan += [(b + gn_name + ".weight", (2 * c,)), (b + gn_name + ".bias", (2 * c,))]
        conv(b + "compress_dim_conv.3", c, 2 * c, 1, 1, True)
        self._plan += [(b + "compress_dim_conv.4.weight", (c,)), (b + "compress_dim_conv.4.bias", (c,))]
        conv(b + "feature_aggreg.0", mid, c, 3, 3, True)
        self._plan += [(b + "feature_aggreg.1.weight", (mid,)), (b + "feature_aggreg.1.bias", (mid,))]
        conv(b + "fc6", spec.BOX_MLP_DIM, mid * p * p, 1, 1, True)
        conv(b + "fc7", spec.BOX_MLP_DIM, spec.BOX_MLP_DIM, 1, 1, True)
        conv(b + "pred", 5 * spec.BOX_NUM_CLASSES, spec.BOX_MLP_DIM, 1, 1, True)

    def _codecs(self):
        """master tensor name -> (reference keys, import(list of reference tensors) -> master-shaped tensor, export(master-
        shaped tensor) -> {reference key or key#part: tensor}).  Conv weights live in [cout][r][s][cin] order; the fused /
        split tensors of this build are assembled from / taken apart into the reference's entries here, in ONE place, for
        state_dict(), named_grads() and the optimizer state alike."""
        h, b = "rpn.head.", "roi_heads.box."
        c, mid, p = spec.FPN_OUT, spec.FPN_OUT // 2, spec.BOX_POOL
        nc = spec.BOX_NUM_CLASSES
        out = {}
        for name, shape in self._plan:
            base, leaf = name.rsplit(".", 1)
            if name == h + "scales":
                keys = ["%sscales.%d.scale" % (h, i) for i in range(5)]
                out[name] = (keys, lambda ts: torch.cat([t.reshape(1) for t in ts]),
                             lambda v, keys=keys: {k: v[i:i + 1] for i, k in enumerate(keys)})
            elif base == h + "cls_ctr":
                keys = [h + "cls_logits." + leaf, h + "centerness." + leaf]
                if leaf == "weight":
                    out[name] = (keys, lambda ts: torch.cat(ts, 0).permute(0, 2, 3, 1),
                                 lambda v, keys=keys: {keys[0]: v[0:1].permute(0, 3, 1, 2), keys[1]: v[1:2].permute(0, 3, 1, 2)})
                else:
                    out[name] = (keys, lambda ts: torch.cat(ts, 0), lambda v, keys=keys: {keys[0]: v[0:1], keys[1]: v[1:2]})
            elif base in (b + "compress_dim_conv.0x", b + "compress_dim_conv.0q"):
                key = b + "compress_dim_conv.0." + leaf
                if leaf == "bias":
                    out[name] = ([key], lambda ts: ts[0], lambda v, key=key: {key: v})
                else:
                    lo = 0 if base.endswith("0x") else c
                    part = "#0" if base.endswith("0x") else "#1"          # merged along the input channels on export
                    out[name] = ([key], lambda ts, lo=lo: ts[0][:, lo:lo + c].permute(0, 2, 3, 1),
                                 lambda v, key=key, part=part: {key + part: v.permute(0, 3, 1, 2)})
            elif base == b + "fc6" and leaf == "weight":
                # Linear over x.view(N, -1) of NCHW maps (box_head.py:151): columns (c, h, w) -> this build's (h, w, c)
                out[name] = ([name], lambda ts: ts[0].view(-1, mid, p, p).permute(0, 2, 3, 1).reshape(-1, 1, 1, mid * p * p),
                             lambda v, name=name: {name: v.reshape(-1, p, p, mid).permute(0, 3, 1, 2).reshape(-1, mid * p * p)})
            elif base == b + "fc7" and leaf == "weight":
                out[name] = ([name], lambda ts: ts[0][:, None, None, :], lambda v, name=name: {name: v.reshape(v.shape[0], -1)})
            elif base == b + "pred":
                keys = [b + "predictor.cls_score." + leaf, b + "predictor.bbox_pred." + leaf]
                if leaf == "weight":
                    out[name] = (keys, lambda ts: torch.cat(ts, 0)[:, None, None, :],
                                 lambda v, keys=keys: {keys[0]: v[:nc].reshape(nc, -1), keys[1]: v[nc:].reshape(4 * nc, -1)})
                else:
                    out[name] = (keys, lambda ts: torch.cat(ts, 0), lambda v, keys=keys: {keys[0]: v[:nc], keys[1]: v[nc:]})
            elif base in self.convs and leaf == "weight":
                out[name] = ([name], lambda ts: ts[0].permute(0, 2, 3, 1), lambda v, name=name: {name: v.permute(0, 3, 1, 2)})
            else:                                       # conv bias, GroupNorm affine
                out[name] = ([name], lambda ts: ts[0], lambda v, name=name: {name: v})
        return out

    def _import_flat(self, flat, ref):
        """Fill a buffer laid out like the masters from reference-named tensors (weights, momentum, ...)."""
        off = 0
        for name, shape in self._plan:
            n = int(math.prod(shape))
            keys, imp, _ = self._codec[name]
            flat[off:off + n].view(shape).copy_(imp([torch.as_tensor(ref[k]).to(self.device, torch.float32) for k in keys]))
            off += n

    def _export_flat(self, flat):
        """Reference-named, reference-shaped copies of a buffer laid out like the masters."""
        out, parts, off = {}, {}, 0
        for name, shape in self._plan:
            n = int(math.prod(shape))
            _, _, exp = self._codec[name]
            for k, v in exp(flat[off:off + n].view(shape)).items():
                v = v.clone(memory_format=torch.contiguous_format)
                if "#" in k:
                    parts.setdefault(k.split("#")[0], {})[int(k.split("#")[1])] = v
                else:
                    out[k] = v
            off += n
        for k, ps in parts.items():
            out[k] = torch.cat([ps[i] for i in sorted(ps)], 1)
        return out

    def _allocate(self, sd):
        total = sum(int(math.prod(s)) for _, s in self._plan)
        total = (total + 63) // 64 * 64
        self.flat_w = torch.zeros(total, device=self.device, dtype=torch.float32)
        self.flat_g = torch.zeros(total, device=self.device, dtype=torch.float32)
        self._codec = self._codecs()
        self._import_flat(self.flat_w, sd)
        off = 0
        for name, shape in self._plan:
            n = int(math.prod(shape))
            wv, gv = self.flat_w[off:off + n].view(shape), self.flat_g[off:off + n].view(shape)
            off += n
            wv.grad = gv
            base, leaf = name.rsplit(".", 1)
            if base in self.convs:
                if leaf == "weight":
                    self.convs[base].w, self.convs[base].gw = wv, gv
                else:
                    self.convs[base].b, self.convs[base].gb = wv, gv
            else:                                                        # GroupNorm affine, Scale scalars
                self.extra[name] = (wv, gv)

    def warm_streams(self):
        """Use every stream of the engine once, in the order they were created: a stream gets its hardware queue at first use,
        and which queue decides what can overtake what.  Call before anything else in the process creates streams of its own
        (torch.distributed / RCCL initialisation); see attach_exchange."""
        cur = torch.cuda.current_stream()
        probe = torch.zeros(64, device=self.device, dtype=torch.float32)
        for st in (cur, self.wstream, self.s1, self.wstream2, self.pstream, self.ustream):
            if st is None:
                continue
            with torch.cuda.stream(st):
                ops.add_mask(probe, None, None, out=probe)
        torch.cuda.synchronize()

    def attach_exchange(self, process_group=None, single_rank=False, wire_dtype=None):
        """(Re)build the gradient exchange on the engine's update stream for a process group initialised AFTER the engine —
        the recommended order: create the engine and run one step (every stream has been used, i.e. has its hardware queue)
        BEFORE torch.distributed / RCCL set up theirs.  Initialising RCCL first shifts which hardware queue each of the
        engine's streams lands on, and the step measured 12-14 % slower (bench.py --live-exchange, same box)."""
        from .dist_utils import GradExchange
        self.join()
        self.pg = process_group
        ranges = [(n, lo, hi) for n, (lo, hi) in self.exchange.ranges.items()]
        self.exchange = GradExchange(self.flat_g, ranges, process_group, single_rank_too=single_rank, wire_dtype=wire_dtype,
                                     comm_stream=self.ustream)
        return self.exchange

    def _bucket_of(self, tensor):
        off = (tensor.data_ptr() - self.flat_w.data_ptr()) // 4
        for name, (lo, hi) in self.exchange.ranges.items():
            if lo <= off < hi:
                return name
        raise AssertionError("tensor outside the flat buffer")

    def _build_pack_tables(self):
        """Allocate ONE flat packed buffer per form (forward, data gradient) and, per gradient bucket, the table that lets a
        single launch repack all of the bucket's convs from the flat fp32 masters (osd_pack_multi)."""
        import numpy as np
        mult = 64 if self.dtype == torch.bfloat16 else 16
        tr = [c for c in self.convs.values() if c.trainable]
        scale_off, scales = {}, []
        off = 0
        for c in tr:
            if c.bn_scale is not None:
                scale_off[c.name] = off
                scales.append(c.bn_scale)
                off += c.cout
        self._flat_scale = torch.cat(scales) if scales else torch.zeros(1, device=self.device)
        base = self.flat_w.data_ptr()
        self._pack = {}
        for form in (0, 1):
            entries, dst_off = [], 0
            for c in tr:
                if form == 0:
                    rows, kpad = ops._round_up(c.cout, 16), ops._round_up(c.cin, mult)
                else:
                    rows, kpad = ops._round_up(c.cin, 16), ops._round_up(c.cout, mult)
                numel = rows * c.r * c.s * kpad
                nb = max(1, min(64, (numel + 256 * 16 - 1) // (256 * 16)))
                entries.append(dict(c=c, src=(c.w.data_ptr() - base) // 4, dst=dst_off, scale=scale_off.get(c.name, -1), rows=rows,
                                    kpad=kpad, nb=nb, numel=numel))
                dst_off += (numel + 63) // 64 * 64
            flat = torch.zeros(dst_off, device=self.device, dtype=self.dtype)
            tables = {}
            for bucket in self.exchange.ranges:
                sub = [e for e in entries if self._bucket_of(e["c"].w) == bucket]
                if not sub:
                    continue
                # 3 int64 offsets + 8 int32 (cout, cin, r, s, rows, kpad, first_block, n_blocks) = 7 x 8 bytes per entry
                tab = np.zeros((len(sub), 7), dtype=np.int64)
                blocks = []
                for i, e in enumerate(sub):
                    c = e["c"]
                    tab[i, 0:3] = (e["src"], e["dst"], e["scale"])
                    tab[i, 3:7] = np.frombuffer(np.array([c.cout, c.cin, c.r, c.s, e["rows"], e["kpad"], len(blocks), e["nb"]],
                                                         dtype=np.int32).tobytes(), dtype=np.int64)
                    blocks += [i] * e["nb"]
                tables[bucket] = dict(table=torch.from_numpy(tab).to(self.device),
                                      blocks=torch.tensor(blocks, dtype=torch.int32, device=self.device), n=len(blocks))
            self._pack[form] = dict(flat=flat, tables=tables)
            if form == 0:
                self._pack_fwd_entry = {id(e["c"]): e for e in entries}      # conv -> its forward-form entry (the fused update)
            for e in entries:
                c = e["c"]
                view = flat[e["dst"]:e["dst"] + e["numel"]].view(e["rows"], c.r, c.s, e["kpad"])
                if form == 0:
                    cout_store = ops._round_up(c.cout, 4)
                    if c.has_bias and c.cout % 16 == 0:
                        bias = c.b                                   # the fp32 master bias IS the epilogue's bias vector
                    else:
                        bias = torch.zeros(ops._round_up(cout_store, 16), device=self.device, dtype=torch.float32)
                        if c.bn_shift is not None:
                            bias[:c.cout] = c.bn_shift
                    c.pc = PackedConv(view, bias, c.cout, cout_store, e["rows"], e["kpad"], c.r, c.s, cin_real=c.cin)
                else:
                    zb = torch.zeros(ops._round_up(c.cin, 16), device=self.device, dtype=torch.float32)
                    c.pd = PackedConv(view, zb, c.cin, c.cin, e["rows"], e["kpad"], c.r, c.s, cin_real=c.cout)
        self._padded_bias = {}
        for c in tr:                            # the two prediction convs keep a padded copy of their 2 / 4 biases
            if c.has_bias and c.cout % 16 != 0:
                self._padded_bias.setdefault(self._bucket_of(c.w), []).append(c)

    def repack(self, buckets=None, forms=(0, 1)):
        """fp32 masters -> kernel-layout weights of the compute dtype: per gradient bucket two launches (forward and
        data-gradient forms).  buckets=None: all of them; forms=(1,): the data-gradient form only (the fused update has
        already written the forward form)."""
        for c in self.convs.values():
            if not c.trainable and c.pc is None:
                c.pc = ops.pack_conv(self._frozen_sd[c.name + ".weight"], bn=None if c.bn_scale is None else tuple(
                    self._frozen_sd[c.name.replace("conv", "bn").replace("downsample.0", "downsample.1") + k]
                    for k in (".weight", ".bias", ".running_mean", ".running_var")), dtype=self.dtype,
                    stem=c.name.endswith("stem.conv1"))
        if not hasattr(self, "_pack"):
            self._build_pack_tables()
        for bucket in (self.exchange.ranges if buckets is None else buckets):
            for form in forms:
                pk = self._pack[form]
                tb = pk["tables"].get(bucket)
                if tb is not None:
                    ops._lib.call("osd_pack_multi", ops._ptr(tb["table"]), ops._ptr(tb["blocks"]), tb["n"], ops._ptr(self.flat_w),
                                  ops._ptr(self._flat_scale), ops._ptr(pk["flat"]), form, ops._dt(pk["flat"]), ops._stream())
            for c in self._padded_bias.get(bucket, ()):
                c.pc.bias[:c.cout] = c.b

    def gn(self, name):
        return self.extra[name + ".weight"], self.extra[name + ".bias"]

    # ------------------------------------------------------------------------------------------------ forward
    BBS = ("backbone.", "supp_backbone.")

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
                if quarter:
                    o2 = ops.conv2d_multi(o1, pcs(p + "conv2"), stride=2, pad=1, act=ACT_RELU)
                    identity = [t[:, ::2, ::2].contiguous() for t in identity]
                    halved = True
                else:
                    o2 = ops.conv2d_multi(o1, pcs(p + "conv2"), pad=1, act=ACT_RELU)
                y = ops.conv2d_multi(o2, pcs(p + "conv3"), act=ACT_RELU, residuals=identity)
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
        p4 = ops.conv2d_multi(inner3, pcs(f + "fpn_layer3"), pad=1)
        inner2 = ops.conv2d_multi(c3, pcs(f + "fpn_inner2"), residuals=inner3, res_mode=RES_UP2X)
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

    TOWERS = ("cls_tower", "bbox_tower")

    def head_forward(self, feats):
        """FCOSHead.forward (fcos.py:83-99).  Layer by layer, BOTH towers over all five levels = ONE conv launch per layer
        (10 pairs: they share the geometry, each tower brings its own weights; level-major order so that the tuner's
        large / small split keeps P3 and P4 of both towers together), then GroupNorm+ReLU of a tower's five levels in two
        launches.  ctx[tower] = ([per layer: (inputs per level, conv outputs per level, ab)], last activations)."""
        if self.towers_merged:
            outs, ctx = self._towers_forward(feats, self.TOWERS)
            return list(zip(outs["cls_tower"], outs["bbox_tower"])), ctx
        # one stream per tower: the HBM-bound GroupNorm passes of one tower run beside the MFMA-bound convs of the other
        main = torch.cuda.current_stream()
        side = self.s1 if self.s1 is not None else main
        if self.split_levels and len(feats) == 5 and None not in (self.s1, self.wstream, self.wstream2):
            # ... and one CHAIN per level group: the levels of a tower never meet before the loss, so P5-P7 (34 pixel tiles,
            # latency-sized launches: 27 us per layer) run as their own conv -> GroupNorm chain on the weight-gradient streams
            # (idle during the forward pass) inside the HBM-bound GroupNorm windows of the P3+P4 chain, instead of in line
            big, small = [0, 1], [2, 3, 4]
            for st in (self.s1, self.wstream, self.wstream2):
                st.wait_stream(main)
            with torch.cuda.stream(self.wstream):
                ocs, ccs = self._towers_forward(feats, ("cls_tower",), small)
            with torch.cuda.stream(self.wstream2):
                obs, cbs = self._towers_forward(feats, ("bbox_tower",), small)
            with torch.cuda.stream(self.s1):
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
        with torch.cuda.stream(side):
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
        if "cls_tower" in towers:
            outs["cls_tower"] = ops.conv2d_grouped(t["cls_tower"], cv[h + "cls_ctr"].pc, pad=1)
        if "bbox_tower" in towers:
            outs["bbox_tower"] = ops.conv2d_grouped(t["bbox_tower"], cv[h + "bbox_pred"].pc, pad=1, act=ACT_EXP_SCALE,
                                                    act_scale_devs=[scales[l:l + 1] for l in lv])
        return outs, {tw: (layers[tw], t[tw]) for tw in towers}

    # ------------------------------------------------------------------------------------------------ loss
    def loss_and_grads(self, head_out, gt_boxes, gt_count):
        """-> losses [4] (cls, reg, centerness, num_pos) on the device, per-level gradients w.r.t. the prediction convs."""
        h = "rpn.head."
        scales, gscales = self.extra[h + "scales"]
        n = head_out[0][0].shape[0]
        sums = torch.zeros(8, device=self.device, dtype=torch.float32)
        nl = len(head_out)
        ops.fcos_loss_levels(0, head_out, gt_boxes, gt_count, spec.FPN_STRIDES[:nl], SIZE_RANGES[:nl], spec.POS_RADIUS,
                             spec.LOSS_GAMMA, spec.LOSS_ALPHA, None, sums)
        gstride = self.convs[h + "bbox_pred"].pd.cin_k
        grads = []
        raw = torch.zeros(5, device=self.device, dtype=torch.float32)
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
        gscales.add_(raw / scales)      # d loss / d scale_l = sum ds * x, x = log(reg) / scale_l
        losses = torch.empty(4, device=self.device, dtype=torch.float32)
        ops._lib.call("osd_fcos_loss_finalize", ops._ptr(sums), ops._ptr(losses), n, ops._stream())
        # {num_pos, sum_w, sum_focal, sum_w*(1-giou), sum_bce}: the un-normalised sums are additive over images (tests)
        self.last_loss_sums = sums
        return losses, grads

    def close(self):
        """Release the ordered-mode scratch buffers registered for this engine's weight-gradient streams."""
        if getattr(self, "ordered_wgrad", False):
            for st in getattr(self, "_wgrad_streams", []):
                try:
                    ops.wgrad_set_workspace(st, 0)
                except Exception:      # noqa: BLE001  (interpreter shutdown)
                    pass
            self.ordered_wgrad = False

    def __del__(self):
        self.close()

    # ------------------------------------------------------------------------------------------------ backward
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
        with torch.cuda.stream(ws):
            fn()
        self._keep.append(tensors)        # keep the operands alive until the side stream has been joined

    def _bucket_ready(self, name, which=0, extra=()):
        """Everything that writes gradient bucket `name` has been enqueued (weight gradients on side stream `which`; for
        the head also the GroupNorm / Scale gradients on the compute stream): start its all-reduce behind those streams
        and, inside train_step, its SGD update + repack behind that.  The update also waits for the current compute
        stream: the bucket's data-gradient convs (enqueued before this point) read the packed weights it rewrites."""
        if not self._overlap or name is None:
            return
        cur = torch.cuda.current_stream()
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
        with torch.cuda.stream(ust):
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

    def _wgrad_grouped(self, c, pairs, which=0):
        self._on_wstream(lambda: ops.conv2d_wgrad_grouped(pairs, c.gw, c.r, c.s, 1, c.r // 2, c.cout, scale=c.bn_scale,
                                                          db=c.gb if c.has_bias else None), pairs, which)

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
            main = torch.cuda.current_stream()
            side = self.s1 if self.s1 is not None else main
            side.wait_stream(main)
            with torch.cuda.stream(side):
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
                if isinstance(ab, list):        # forward ran one chain per level group: one saved-statistics block each
                    dus[tw], lo = [], 0
                    for ab_g in ab:
                        k = ab_g.shape[0]
                        dus[tw] += ops.groupnorm_relu_bwd_levels(u[lo:lo + k], d_t[tw][lo:lo + k], ab_g, gw, gbeta, ggw, ggb, spec.GN_GROUPS)
                        lo += k
                else:
                    dus[tw] = ops.groupnorm_relu_bwd_levels(u, d_t[tw], ab, gw, gbeta, ggw, ggb, spec.GN_GROUPS,
                                                            ws=gws[(tw, i)] if fuse else None, fused_mask=fused[tw])
                items[tw] += [(t_in[l], dus[tw][l], c.gw, c.bn_scale, c.gb if c.has_bias else None) for l in range(nl)]
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
            self._on_wstream(lambda it=items[tw], c0=c0: ops.conv2d_wgrad_multi(it, c0.r, c0.s, 1, c0.r // 2, c0.cout),
                             items[tw], self.TOWERS.index(tw))
        return d_t

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
        return None

    # ------------------------------------------------------------------------------------------------ second stage
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

    # ------------------------------------------------------------------------------------------------ step
    def forward_backward(self, images, queries, gt_boxes, gt_count, with_proposals=True, image_sizes=None):
        """One training forward + backward.  images [B,3,H,W], queries [B*S,3,h,w] fp32 NCHW on the device;
        gt_boxes [B, G, 4] fp32 xyxy, gt_count [B] int32.  Returns losses [4] = (cls, reg, centerness, num_pos)."""
        from . import model
        from .layers import ImageList
        # padded batches as the collator hands them over (layers.ImageList, or transforms.collate(..., stem_dtype) =
        # ops.PackedImages: already in the stem conv's input format): every image's / query's true size travels along
        query_sizes = None
        if isinstance(images, ImageList):
            images, image_sizes = images.tensors, images.image_sizes
        elif isinstance(images, ops.PackedImages) and image_sizes is None:
            image_sizes = images.image_sizes
        if isinstance(queries, ImageList):
            queries, query_sizes = queries.tensors, queries.image_sizes
        elif isinstance(queries, ops.PackedImages):
            query_sizes = queries.image_sizes
        main, s1 = torch.cuda.current_stream(), self.s1
        # train_step(defer_join) left the previous step's tail (last weight gradients, exchange, update, repack, proposals)
        # running on the side streams: the frozen prefix of this forward goes first, then the main stream joins them
        deferred, self._deferred = self._deferred, None
        prev_keep, self._keep = self._keep, []

        def join_previous():
            if deferred is not None:
                for ev in deferred["events"]:       # events recorded when the previous step returned
                    main.wait_event(ev)
            # optimizer.zero_grad() (engine/trainer.py:89): the weight-gradient kernels accumulate.  A step whose every bucket went
            # through the fused update has had its gradients zeroed by that kernel, behind its read (consume_grads)
            if not self._grads_clean:
                self.flat_g.zero_()
            self._grads_clean = False
            self.exchange.begin()
        if deferred is None:
            join_previous()
        batch = images.shape[0]
        shots = queries.shape[0] // batch
        q_sizes = [tuple(queries.shape[-2:])] * queries.shape[0] if query_sizes is None else [tuple(v) for v in query_sizes]
        rois = model.whole_image_rois(q_sizes, self.device)
        # ---- forward: both backbones in lockstep (one launch per layer), query pooling, correlation, head
        lock = self.lockstep
        side = s1 if s1 is not None else main
        af = join_previous if deferred is not None else None

        def pool(qf):
            out = []
            for feat, scale in zip(qf, spec.POOLER_SCALES):
                v = ops.roi_align(feat, rois, scale, 1, 1, spec.POOLER_SAMPLING_RATIO)
                out.append(ops.shot_mean(v.view(v.shape[0], -1), batch))
            return out
        if lock:
            (feats, qfeats), (tctx, qctx) = self.backbones_forward(images, queries, after_frozen=af)
            pooled = pool(qfeats)
        else:       # the query backbone + pooling on the second stream beside the target backbone
            if deferred is not None and s1 is not None:
                for ev in deferred["events"]:
                    s1.wait_event(ev)
            side.wait_stream(main)
            if s1 is None and af is not None:       # single stream: the join cannot be deferred past the query backbone
                af()
                af = None
            with torch.cuda.stream(side):
                (qfeats,), (qctx,) = self._backbones_forward(self.BBS[1:], (queries,))
                pooled = pool(qfeats)
            (feats,), (tctx,) = self._backbones_forward(self.BBS[:1], (images,), af)
            main.wait_stream(side)
        prev_keep = deferred = self._joined_refs = None   # what the previous step's side work reads is released only now
        combined = ops.correlate_levels(feats, pooled) if self.corr_levels else [ops.correlate(f, q) for f, q in zip(feats, pooled)]
        head_out, hctx = self.head_forward(combined)
        if with_proposals:      # box_selector_train under no_grad (fcos.py:196-199): proposals for the second stage,
            ps = self.pstream if self.pstream is not None else main                        # independent of loss/backward
            if ps is not main:
                ps.wait_stream(main)
            with torch.cuda.stream(ps):
                pb, ps_, pc = model.run_proposals(head_out, images.shape[-2], images.shape[-1],
                                                  spec.PRE_NMS_TOP_N_TRAIN, spec.POST_NMS_TOP_N_TRAIN, spec.NMS_THRESH,
                                                  image_sizes=image_sizes,    # padded batch: clip to each image's size
                                                  depth=self._prop_depth)     # lagged hint: how deep NMS has to read
                # add_gt_proposals (fcos/inference.py:139-160,279): the ground-truth boxes join the training proposals
                self.proposals = ops.append_gt_boxes(pb, ps_, pc, gt_boxes, gt_count)
                if self.second_stage:       # roi_heads on the plain target / query features (generalized_rcnn.py:317), beside
                    box_out = self.box_head_forward_backward(feats, qfeats, q_sizes, shots, self.proposals, gt_boxes, gt_count,
                                                             keys=self.box_keys)    # the first stage's loss and backward
                    self._bucket_ready("box_head", 0, [ps])
        # ---- loss + backward
        losses, pred_grads = self.loss_and_grads(head_out, gt_boxes, gt_count)
        self.last_head_out, self.last_pred_grads = head_out, pred_grads      # (tests: conditioning of the Scale gradients)
        d_comb = self.head_backward(combined, hctx, pred_grads)
        self._bucket_ready("head", 0, [st for st in (main, self.wstream, self.wstream2) if st is not None])
        # correlation backward (generalized_rcnn.py:307-311): d q = sum_hw g * feat, d feat = g * q
        dq = ops.correlate_bwd_query_levels(d_comb, feats) if self.corr_levels else \
            [ops.correlate_bwd_query(g, f) for g, f in zip(d_comb, feats)]
        second = self.second_stage and with_proposals
        ps = self.pstream if self.pstream is not None else main
        if s1 is not None:
            s1.wait_stream(main)
            if second and ps is not main:
                s1.wait_stream(ps)
        if second:
            if ps is not main:
                main.wait_stream(ps)
            self.box_losses, gx, gqs = box_out
        with torch.cuda.stream(side):      # the query branch's small pooling-backward chain beside d feat
            dQ = []
            for dql, qf, scale in zip(dq, qfeats, spec.POOLER_SCALES):
                dv = ops.shot_mean_bwd(dql, shots)
                gxq = ops.roi_align_bwd(dv.view(-1, 1, 1, dv.shape[-1]), rois, qf.shape, scale, 1, 1,
                                        spec.POOLER_SAMPLING_RATIO)
                dQ.append(ops.cast_f32(gxq, self.dtype))
            if second:                  # the second stage's gradient w.r.t. the query features joins the first stage's
                for lvl, gq in gqs:
                    if shots > 1:       # only the first query of every image reached the second stage's loss
                        full = torch.zeros((gq.shape[0] * shots,) + tuple(gq.shape[1:]), device=self.device, dtype=torch.float32)
                        full[::shots] = gq
                        gq = full
                    dQ[lvl] = ops.add_mask(dQ[lvl], ops.cast_f32(gq, self.dtype))
            if not lock:
                self.backbones_backward([qctx], [dQ], which0=1)
        dP = ops.correlate_levels(d_comb, pooled) if self.corr_levels else [ops.correlate(g, q) for g, q in zip(d_comb, pooled)]
        if second:                      # ... and w.r.t. the target FPN outputs (generalized_rcnn.py:317: the plain features)
            dP = [ops.add_mask(d, ops.cast_f32(g, self.dtype)) for d, g in zip(dP, gx)]
            self._keep.append((gx, gqs))
        if lock:
            if s1 is not None:
                main.wait_stream(s1)
            self.backbones_backward([tctx, qctx], [dP, dQ])
        else:
            self.backbones_backward([tctx], [dP])
        self._keep.append((dq, dP, dQ, d_comb, pred_grads))
        if self._defer_now:
            # leave the tail on the side streams; the next forward_backward (or join()) orders the main stream after them
            events = []
            for st in (s1, self.wstream, self.wstream2, self.pstream, self.ustream, self.exchange.comm):
                if st is not None:
                    ev = torch.cuda.Event()
                    ev.record(st)
                    events.append(ev)
            self._deferred = dict(events=events, refs=(qctx, tctx, hctx, feats, qfeats, pooled, combined, head_out, dQ, rois))
            return losses
        if s1 is not None:
            main.wait_stream(s1)
        if self.wstream is not None:
            main.wait_stream(self.wstream)
        if self.wstream2 is not None:
            main.wait_stream(self.wstream2)
        if with_proposals and self.pstream is not None:
            main.wait_stream(self.pstream)
        return losses

    def join(self):
        """Order the current stream after everything a deferred train_step left on the side streams (weight gradients,
        exchange, update, repack, proposals).  Call before reading weights, gradients or proposals on this stream."""
        d, self._deferred = self._deferred, None
        if d is not None:
            cur = torch.cuda.current_stream()
            for ev in d["events"]:
                cur.wait_event(ev)
            self._joined_refs = d["refs"]          # dropped at the next step: the waits above are enqueued, not finished

    def reduce_gradients(self):
        """DDP gradient averaging (tools/train_net.py:83-88).  The buckets of the flat fp32 buffer were handed to RCCL as
        they became final during backward (dist_utils.GradExchange); this waits for them."""
        self.exchange.finish()        # buckets not announced during backward (graph replay, single stream) go now

    def _build_sgd_table(self, weights, biases):
        """Per gradient bucket the table of osd_sgd_momentum_pack_multi (one launch updates every tensor of the bucket AND writes
        the forward-form packed weights of its conv tensors; OSD_NO_FUSED_REPACK=1: osd_sgd_momentum_multi + the two-form repack)."""
        import numpy as np
        base = self.flat_w.data_ptr()
        conv_of = {c.w.data_ptr(): c for c in self.convs.values() if c.trainable}
        rows = {name: [] for name in self.exchange.ranges}
        for group, lr_mult, wd in ((weights, 1.0, self.weight_decay), (biases, 2.0, 0.0)):
            for t in group:
                rows[self._bucket_of(t)].append(((t.data_ptr() - base) // 4, t.numel(), lr_mult, wd, conv_of.get(t.data_ptr())))
        fuse = os.environ.get("OSD_NO_FUSED_REPACK", "0") == "0"
        tables = {}
        for name, rs in rows.items():
            if not rs:
                continue
            fused = fuse
            tab = np.zeros((len(rs), 8 if fused else 4), dtype=np.int64)         # 64 / 32 bytes per entry
            blocks = []
            for i, (off, n, lm, wd, c) in enumerate(rs):
                nb = max(1, min(64, (n + 256 * 16 - 1) // (256 * 16)))
                tab[i, 0], tab[i, 1] = off, n
                tab[i, 2] = np.frombuffer(np.array([lm, wd], dtype=np.float32).tobytes(), dtype=np.int64)[0]
                tab[i, 3] = np.frombuffer(np.array([len(blocks), nb], dtype=np.int32).tobytes(), dtype=np.int64)[0]
                if fused:
                    tab[i, 4] = tab[i, 5] = -1
                    if c is not None:
                        e = self._pack_fwd_entry[id(c)]
                        tab[i, 4], tab[i, 5] = e["dst"], e["scale"]
                        tab[i, 6:8] = np.frombuffer(np.array([c.cin, c.r * c.s, e["kpad"], 0], dtype=np.int32).tobytes(), dtype=np.int64)
                blocks += [i] * nb
            tables[name] = dict(table=torch.from_numpy(tab).to(self.device), fused=fused,
                                blocks=torch.tensor(blocks, dtype=torch.int32, device=self.device), n=len(blocks))
        self._sgd = dict(tables=tables, buf=torch.zeros_like(self.flat_w), steps=0)

    def _update_bucket(self, name):
        """SGD(momentum) on the bucket's masters, then its repack, on the current stream."""
        sg = self._sgd
        tb = sg["tables"].get(name)
        fused = tb is not None and tb["fused"]
        if fused:
            pk = self._pack[0]["flat"]
            ops._lib.call("osd_sgd_momentum_pack_multi", ops._ptr(tb["table"]), ops._ptr(tb["blocks"]), tb["n"],
                          ops._ptr(self.flat_w), ops._ptr(self.flat_g), ops._ptr(sg["buf"]), ops._ptr(self._flat_scale), ops._ptr(pk),
                          ops._dt(pk), float(self.lr), float(self.momentum), int(sg["steps"] == 0), int(self.consume_grads), ops._stream())
            if self.consume_grads:
                self._zeroed.add(name)
        elif tb is not None:
            ops._lib.call("osd_sgd_momentum_multi", ops._ptr(tb["table"]), ops._ptr(tb["blocks"]), tb["n"],
                          ops._ptr(self.flat_w), ops._ptr(self.flat_g), ops._ptr(sg["buf"]), float(self.lr),
                          float(self.momentum), int(sg["steps"] == 0), ops._stream())
        self.repack([name], forms=(1,) if fused else (0, 1))      # (the padded copies of the 2 / 4 prediction biases ride along)
        self._updated.add(name)

    def optimizer_step(self):
        """Apply the update to every bucket train_step has not already updated behind the backward pass."""
        if self.opt is not None:
            self.opt.step()
            self.repack()
            return
        if self._overlap:              # (never inside a captured graph: capture() turns the overlap off)
            main = torch.cuda.current_stream()
            main.wait_stream(self.ustream)
            if self.exchange.comm is not None:
                main.wait_stream(self.exchange.comm)
        for name in self.exchange.ranges:
            if name not in self._updated:
                self._update_bucket(name)
        self._end_of_update()
        self._sgd["steps"] += 1

    def _end_of_update(self):
        self._updated = set()
        self._grads_clean = bool(self.consume_grads and self._zeroed >= set(self._sgd["tables"]))      # every bucket consumed
        self._zeroed = set()

    def train_step(self, images, queries, gt_boxes, gt_count):
        """forward + loss + backward + gradient averaging + SGD + repack.  With the fused optimiser each bucket's exchange,
        update and repack run on a side stream as soon as the bucket is final, beside the rest of the backward pass."""
        self._fuse_update = self.opt is None and self._overlap
        # defer_join: do not make the main stream wait for the step's tail (the last stage's weight gradients, their
        # exchange, update and repack): the next step's frozen layers (stem, layer1) run beside it.  Every bucket's update
        # already sits on the update / communication stream in fused mode, so nothing is left to enqueue here
        self._defer_now = bool(self.defer_join and self._fuse_update and self.wstream is not None)
        try:
            losses = self.forward_backward(images, queries, gt_boxes, gt_count)
        finally:
            self._fuse_update = False
            deferred_step, self._defer_now = self._defer_now, False
        if deferred_step and all(name in self._updated for name in self.exchange.ranges):
            self._end_of_update()
            self._sgd["steps"] += 1
            return losses
        self.join()
        self.reduce_gradients()
        self.optimizer_step()
        return losses

    def capture(self, images, queries, gt_boxes, gt_count, warmup=2):
        """Capture the training step into hipGraphs (forward+loss+backward as one graph, SGD + repack as another, with the
        RCCL all-reduce between them launched normally): ~1500 launches per step become two graph launches.  The learning
        rate is baked in at capture time; call capture() again after changing it."""
        # static copies of the inputs; a padded batch (layers.ImageList / ops.PackedImages) keeps its per-image sizes, which
        # are baked into the captured graph together with the batch shape
        self._static = [_static_clone(t) for t in (images, queries, gt_boxes, gt_count)]
        self._overlap = False          # no collectives inside a captured graph: the exchange runs between the two graphs
        self.join()
        self.defer_join = False        # a captured graph must join every stream it forked
        self._prop_depth = None        # the lagged NMS-depth feedback queries events from the host: not capturable
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self.train_step(*self._static)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self._g_fb = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self._g_fb):
            self._static_losses = self.forward_backward(*self._static)
        self._g_opt = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self._g_opt):
            self.optimizer_step()
        return self

    def replay_step(self, images=None, queries=None, gt_boxes=None, gt_count=None):
        for dst, src in zip(self._static, (images, queries, gt_boxes, gt_count)):
            if src is not None:
                _static_copy(dst, src)
        self._g_fb.replay()
        self.reduce_gradients()
        self._g_opt.replay()
        return self._static_losses

    # ------------------------------------------------------------------------------------------------ state
    def state_dict(self):
        """The reference's state_dict (same names, OIHW shapes) with the current fp32 master weights: what
        `DetectronCheckpointer.save` (utils/checkpoint.py:35-52) would write for the hot-path modules.  Frozen tensors
        (stem, layer1, every FrozenBN buffer) are returned unchanged."""
        self.join()
        out = {k: v.clone() for k, v in self._frozen_sd.items()}
        out.update(self._export_flat(self.flat_w))
        return out

    def named_grads(self):
        """Reference-named gradients (OIHW) for parity tests."""
        return self._export_flat(self.flat_g)

    # ------------------------------------------------------------------------------------------------ optimiser state
    def optimizer_state_dict(self):
        """What the reference checkpoint stores under 'optimizer' (utils/checkpoint.py:42-46: torch.optim.SGD.state_dict()),
        keyed by reference parameter NAME instead of torch's positional ids: momentum buffers (OIHW, reference names), the
        number of steps taken (the first step initialises the buffer with the gradient, torch.optim.SGD semantics) and
        the hyper-parameters.  load_optimizer_state_dict() restores it, so a resumed run continues with its momentum."""
        self.join()
        if self.opt is not None:
            raise NotImplementedError("optimizer='torch' (A/B mode): use self.opt.state_dict()")
        return {"momentum_buffer": self._export_flat(self._sgd["buf"]), "steps": int(self._sgd["steps"]),
                "lr": float(self.lr), "momentum": float(self.momentum), "weight_decay": float(self.weight_decay)}

    def load_optimizer_state_dict(self, state):
        self.join()
        if self.opt is not None:
            raise NotImplementedError("optimizer='torch' (A/B mode): use self.opt.load_state_dict()")
        if float(state.get("weight_decay", self.weight_decay)) != self.weight_decay:
            raise ValueError("weight decay is baked into the update tables: construct the engine with weight_decay=%r"
                             % state["weight_decay"])
        self._import_flat(self._sgd["buf"], state["momentum_buffer"])
        self._sgd["steps"] = int(state["steps"])
        self.lr = float(state.get("lr", self.lr))
        self.momentum = float(state.get("momentum", self.momentum))
