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

from . import streams

from . import ops, spec

from .train_backward import BackwardPass
from .train_forward import SIZE_RANGES, ForwardPass      # noqa: F401
from .train_second_stage import SecondStage
from .train_state import State
from .train_update import Update


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


class TrainEngine(ForwardPass, BackwardPass, SecondStage, Update, State):
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
        # A/B switches (measured on one box, tools/ab_round4.sh): both backbones per launch / both towers per launch
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
        # A/B only (OSD_STREAM_ALIAS="u=w,p=w2"): let one role run on another role's stream — fewer streams for the 4 hardware queues
        for pair in [a for a in os.environ.get("OSD_STREAM_ALIAS", "").split(",") if "=" in a]:
            names = {"w": "wstream", "w2": "wstream2", "s1": "s1", "p": "pstream", "u": "ustream"}
            dst, src = pair.split("=")
            if getattr(self, names[src]) is not None:
                setattr(self, names[dst], getattr(self, names[src]))
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
        # the prediction convs' data gradient as a 1x1 conv over the gathered dy matrix the weight gradient builds anyway (round 5;
        # OSD_NO_PRED_DGRAD_GEMM=1: the 3x3 conv over dy, A/B)
        self.pred_dgrad_gemm = os.environ.get("OSD_NO_PRED_DGRAD_GEMM", "0") == "0"
        self._pred_dgrad = {}
        # experiment (train_backward._release_held_wgrads): "" = the towers' weight gradients go out where the head backward ends
        self.tower_wgrad_at = os.environ.get("OSD_TOWER_WGRAD_AT", "")
        self._held_wgrads = []
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
        # the FPN's P3 and P4 output convs (and their data gradients) as one launch each (OSD_NO_FPN_GROUPED=1: one launch per level)
        self.fpn_out_grouped = os.environ.get("OSD_NO_FPN_GROUPED", "0") == "0"
        # the tower convs' bias gradients from the GroupNorm backward's sums instead of the weight-gradient launch's column sums
        # (-11 % on that launch); not in ordered mode (its sum order over (level, image) is atomic) and bf16 / fp32 alike
        # (a conv-bias GroupNorm backward always takes the two-launch kernels, so asking for the one-pass BACKWARD —
        # OSD_GN_ONEPASS=b / 1, an A/B configuration — turns the conv-bias form off; otherwise that switch could not reach its kernel)
        self.gn_conv_db = (not self.ordered_wgrad and os.environ.get("OSD_NO_GN_CONV_DB", "0") == "0" and not ops.GN_ONEPASS_BWD)
        self.fuse_gn_fwd = (not self.ordered_wgrad and self.dtype == torch.bfloat16 and os.environ.get("OSD_GN_FWD_FUSION", "0") != "0")
        self._gnf_ws = {}
        # torch hands out stream handles from a pool, so a handle may carry an earlier engine's registration: set the mode of
        # this engine's weight-gradient streams explicitly either way (close() / __del__ release the scratch buffers)
        # (the second stage's weight gradients run on the proposal stream: box_head_forward_backward)
        cands = (self.wstream, self.wstream2) + ((self.pstream,) if self.second_stage else ())
        self._wgrad_streams = list({id(x): x for x in cands if x is not None}.values()) or [streams.current()]
        if self.ordered_wgrad and self.second_stage:
            import warnings
            warnings.warn("ordered_wgrad makes every conv_wgrad launch bit-reproducible, the second stage's included; its "
                          "ROI-pool backward (osd_roi_pool_levels_bwd) still scatters with fp32 atomics, so the gradients it "
                          "feeds into both backbones are reproducible only up to the order of those adds")
        for st in self._wgrad_streams:
            ops.wgrad_set_workspace(st, (1 << 30) if self.ordered_wgrad else 0)
        # which streams share a hardware queue is decided by the order of their first use: fix it now (see warm_streams)
        if self.wstream is not None:
            self.warm_streams()
        self.defer_join = False       # opt-in: train_step leaves its tail on the side streams (see train_step / join)
        self._deferred, self._defer_now, self._joined_refs = None, False, None
        # (Round 6, measured and removed: leaving the proposal stream out of the events the next step waits for — its NMS is data-dependent
        # and reads every candidate once a model's scores have sharpened.  11.62 vs 11.67 ms per step in that regime
        # (profiles/r6_sustained.txt): the main chain does not WAIT for the proposals, it shares the chip with them; what helped is the
        # deep-regime path of osd_proposals_sort_nms_hint, 14.2 -> 11.6 ms.)
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
        """Use every stream of the engine once, in a chosen order: ROCm multiplexes a process's streams onto 4 hardware queues, a
        queue executes its packets in order, and which two of the six streams end up sharing one is decided at first use.  Measured
        (rocprofv3 queue ids, tools/queue_map.py): main, the two weight-gradient streams and s1 (query branch + bbox tower) always
        get queues 1 / 3 / 2 / 4; of the proposal stream and the update stream, the one used FIRST shares queue 4 with s1, the other
        queue 3 with a weight-gradient stream.  In the natural order of a first step the proposals come first — and then the bbox
        tower's backward chain (s1) sits behind the 0.75 ms NMS pipeline launched at the loss, the main stream idles ~0.8 ms at the
        point where it needs the bbox tower's gradient, and the GroupNorm-under-conv overlap of the two towers is lost.  With the
        update stream used before the proposal stream (default order below) the proposals share a queue with weight gradients,
        which have slack: +1.2 % .. +3.5 % images/s on three boxes (profiles/r4_stream_order_ab.txt).  OSD_WARM_ORDER=main,w,s1,w2,p,u
        restores the old pairing.  The engine calls this at construction; call it again before anything else in the process
        creates streams of its own only if the engine was built without it (torch.distributed / RCCL: see attach_exchange)."""
        cur = streams.current()
        probe = torch.zeros(64, device=self.device, dtype=torch.float32)
        by_name = {"main": cur, "w": self.wstream, "s1": self.s1, "w2": self.wstream2, "p": self.pstream, "u": self.ustream}
        order = [n for n in os.environ.get("OSD_WARM_ORDER", "main,s1,u,w,w2,p").split(",") if n in by_name]
        for name in order:
            st = by_name[name]
            if st is None:
                continue
            with streams.on(st):
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

    def gn(self, name):
        return self.extra[name + ".weight"], self.extra[name + ".bias"]

    # ------------------------------------------------------------------------------------------------ forward
    BBS = ("backbone.", "supp_backbone.")

    TOWERS = ("cls_tower", "bbox_tower")

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
        main, s1 = streams.current(), self.s1
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
            if ops.QUERY_POOL_LEVELS and len(qf) <= 8:      # all levels in one launch (round 6: 10 launches -> 1)
                return ops.query_pool_levels(qf, rois, spec.POOLER_SCALES, batch, spec.POOLER_SAMPLING_RATIO)
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
            with streams.on(side):
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
            with streams.on(ps):
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
        if not self._held_wgrads:
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
        with streams.on(side):      # the query branch's small pooling-backward chain beside d feat
            dQ = []
            if ops.QUERY_POOL_LEVELS and len(dq) <= 8:      # all levels in three launches (round 6: 20 launches + 5 memsets before)
                dQ = ops.query_pool_levels_bwd(dq, rois, [tuple(qf.shape) for qf in qfeats], spec.POOLER_SCALES, shots,
                                               spec.POOLER_SAMPLING_RATIO, self.dtype)
            else:
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
            cur = streams.current()
            for ev in d["events"]:
                cur.wait_event(ev)
            self._joined_refs = d["refs"]          # dropped at the next step: the waits above are enqueued, not finished

    def reduce_gradients(self):
        """DDP gradient averaging (tools/train_net.py:83-88).  The buckets of the flat fp32 buffer were handed to RCCL as
        they became final during backward (dist_utils.GradExchange); this waits for them."""
        self.exchange.finish()        # buckets not announced during backward (graph replay, single stream) go now

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
        side.wait_stream(streams.current())
        with streams.on(side):
            for _ in range(warmup):
                self.train_step(*self._static)
        streams.current().wait_stream(side)
        torch.cuda.synchronize()
        # the captured forward + backward owns its own zeroing of the gradient buffer: `_grads_clean` is a HOST flag, and a graph
        # captured while it was set would hold no memset — correct only while every replay follows a consuming update.  An eager
        # forward_backward() between two replays, or a test writing flat_g, would then be accumulated on top of
        self._grads_clean = False
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
        # what the host flags say after a replay is what they said at capture time; be conservative for whatever eager call
        # follows: it zeroes the buffer itself (one memset), and the next replay zeroes inside its graph
        self._grads_clean, self._zeroed, self._updated = False, set(), set()
        return self._static_losses
