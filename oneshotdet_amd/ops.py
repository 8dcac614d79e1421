"""Thin torch-tensor wrappers over the C-ABI (include/oneshotdet_hip.h).  torch supplies device memory and the stream;
all arithmetic happens in liboneshotdet_hip.so.  Activations are NHWC tensors [N, H, W, C] (float32 or bfloat16)."""
import ctypes as C
import os

import torch

from . import _lib
from ._lib import (ACT_EXP_SCALE, ACT_NONE, ACT_RELU, GN_SPLITS, OSD_BF16, OSD_F32, RES_DOWN2X, RES_NONE, RES_SAME, RES_UP2X,
                   ConvDesc)

__all__ = ["ACT_NONE", "ACT_RELU", "ACT_EXP_SCALE", "RES_NONE", "RES_SAME", "RES_UP2X", "RES_DOWN2X"]


from . import streams

# the current stream's raw handle without torch.cuda.current_stream()'s ~8 us of Python per call (streams.py)
_stream_handle = streams.raw_handle


def _stream():
    return C.c_void_p(_stream_handle())


from .trace import rec as _rec      # launch trace (tests only): oneshotdet_amd/trace.py


def _dt(t):
    if t.dtype == torch.float32:
        return OSD_F32
    if t.dtype == torch.bfloat16:
        return OSD_BF16
    raise TypeError("unsupported activation dtype %s" % t.dtype)


class _Ptr(C.c_void_p):
    """a device address that keeps its tensor alive: `_ptr(x.contiguous())` on a non-contiguous x creates a temporary whose block the
    caching allocator would hand to the NEXT allocation made while the argument list is still being built (before the launch)"""


def _ptr(t):
    if t is None:
        return C.c_void_p(0)
    p = _Ptr(t.data_ptr())
    p.keep = t
    return p


def _chk_dev(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _lib.OsdError("oneshotdet_amd ops need device tensors (no CPU path): got %s" % t.device)


def conv_out(n, k, s, p):
    return (n + 2 * p - k) // s + 1


class PackedConv(object):
    """Weights of one conv in kernel layout: [w_rows][R][S][cin_k] rows (K contiguous), fp32 bias (BN folded)."""
    __slots__ = ("w", "bias", "cout", "cout_store", "w_rows", "cin_k", "cin_real", "r", "s", "stem")

    def __init__(self, w, bias, cout, cout_store, w_rows, cin_k, r, s, stem=False, cin_real=None):
        self.w, self.bias, self.cout, self.cout_store = w, bias, cout, cout_store
        self.w_rows, self.cin_k, self.r, self.s, self.stem = w_rows, cin_k, r, s, stem
        self.cin_real = cin_k if cin_real is None else cin_real


def _round_up(x, m):
    return (x + m - 1) // m * m


def pack_conv(weight, bias=None, bn=None, dtype=torch.float32, stem=False):
    """weight: OIHW fp32 device tensor.  bn = (weight, bias, running_mean, running_var) of a FrozenBatchNorm2d
    (layers/batch_norm.py:19-24, NO eps) folded as w' = w * scale, b' = bn_bias - mean * scale."""
    _chk_dev(weight)
    cout, cin, r, s = weight.shape
    weight = weight.contiguous().float()
    scale = None
    if bn is not None:
        g, b, mean, var = (t.float() for t in bn)
        scale = (g * var.rsqrt()).contiguous()
        fbias = b - mean * scale
    else:
        fbias = bias.float() if bias is not None else torch.zeros(cout, device=weight.device)
    cout_store = _round_up(cout, 4)
    w_rows = _round_up(cout, 16)
    bias_p = torch.zeros(_round_up(cout_store, 16), device=weight.device, dtype=torch.float32)
    bias_p[:cout] = fbias
    if stem:
        assert (cin, r, s) == (3, 7, 7)
        wp = torch.empty((w_rows, 7, 32), device=weight.device, dtype=dtype)
        _lib.call("osd_pack_stem_weight", _p(weight), _p(scale), _p(wp), cout, w_rows, _dt(wp), _stream())
        return PackedConv(wp, bias_p, cout, cout_store, w_rows, 32, 7, 1, stem=True, cin_real=3)
    mult = 64 if dtype == torch.bfloat16 else 16
    cin_pad = _round_up(cin, mult)
    wp = torch.empty((w_rows, r, s, cin_pad), device=weight.device, dtype=dtype)
    _lib.call("osd_pack_conv_weight", _p(weight), _p(scale), _p(wp), cout, cin, r, s, w_rows, cin_pad, _dt(wp),
              _stream())
    return PackedConv(wp, bias_p, cout, cout_store, w_rows, cin_pad, r, s, cin_real=cin)


def pack_image(images, dtype, hp, wp, pad_t=3, pad_l=3):
    """NCHW fp32 [N,3,H,W] -> zero padded NHWC4 [N,hp,wp,4] (the stem conv's padding=3 materialised)."""
    _chk_dev(images)
    n, c, h, w = images.shape
    assert c == 3 and images.dtype == torch.float32
    out = torch.empty((n, hp, wp, 4), device=images.device, dtype=dtype)
    _lib.call("osd_pack_image", _ptr(images.contiguous()), _p(out), n, h, w, hp, wp, pad_t, pad_l, _dt(out), _stream())
    _rec("pack_image", x=images, out=out, pad_t=pad_t, pad_l=pad_l)
    return out


class PackedImages(object):
    """A batch already in the stem conv's input format (what pack_image produces): `tensor` [N, hp, wp, 4] zero padded NHWC4
    with every image at (3, 3), `hw` = (H, W) of the (padded) batch, `image_sizes` = every image's true (h, w).  Written
    directly by transforms.collate(..., stem_dtype=...), so the float NCHW batch is never materialised."""

    def __init__(self, tensor, hw, image_sizes):
        self.tensor, self.hw, self.image_sizes = tensor, tuple(hw), [tuple(s) for s in image_sizes]

    @property
    def shape(self):
        return (self.tensor.shape[0], 3, self.hw[0], self.hw[1])


def stem_input(images, dtype):
    """images: NCHW fp32 [N,3,H,W] or PackedImages -> (padded NHWC4 stem input, (Ho, Wo) of the 7x7/2 stem conv)."""
    n, _, h, w = images.shape
    ho, wo = conv_out(h, 7, 2, 3), conv_out(w, 7, 2, 3)
    if isinstance(images, PackedImages):
        assert images.tensor.dtype == dtype, "PackedImages were written as %s, the engine computes in %s" % (images.tensor.dtype, dtype)
        return images.tensor, (ho, wo)
    hp, wp = max(2 * (ho - 1) + 7, h + 3), max(2 * (wo - 1) + 8, w + 3)
    wp += wp & 1
    return pack_image(images, dtype, hp, wp), (ho, wo)


from .tuner import (TUNE_LOG, refine_in_step, wgrad_timer, ALGO_CACHE, CONV_ALGO_PX, CONV_ALGO_PX_WIDE, SPLIT_CACHE, WGRAD_ALGO_CACHE, _TUNING, _candidate_runs, _photo_finish, _time_launches, _tune, _tune_wgrad,      # noqa: E402,F401
                    conv_algo_candidates, replaying, tuning, wgrad_algo_candidates, wgrad_xr_candidates)


class ConvSrc2(C.Structure):
    """osd_conv_src2: the second pixel source of a 1x1 conv."""
    _fields_ = [("x", C.c_void_p), ("w2", C.c_void_p), ("cin2", C.c_int32), ("h", C.c_int32), ("w", C.c_int32),
                ("stride", C.c_int32)]


def conv2d(x, pc, stride=1, pad=0, act=ACT_NONE, res=None, res_mode=RES_NONE, relu_in=False, act_scale=1.0, out=None,
           out_hw=None, algo=None, mask=None, act_scale_dev=None, x2=None, x2_stride=1, pc2=None, bias=None):
    """x NHWC [N,H,W,C] -> [N,Ho,Wo,cout_store].  For the stem, x is the padded NHWC4 image from pack_image and
    out_hw gives (Ho, Wo).  x2 (1x1 convs): a second NHWC source whose channels follow x's in the packed K (pc packed from
    cat([W_x, W_x2], dim=1)); output pixel (ho, wo) reads x2[ho * x2_stride, wo * x2_stride].  With pc2 the two parts keep
    their own packed weights (pc for x, pc2 for x2: two separately trained convs as one GEMM) and `bias` is the epilogue's
    bias vector (the sum of both convs' folded shifts)."""
    _chk_dev(x, res, out)
    if x2 is not None:
        return _conv2d_two_sources(x, x2, x2_stride, pc, act, out, algo, pc2, bias)
    # The descriptor of a (layer geometry, input shape, epilogue) combination never changes: it is built once and reused — ~25 ctypes
    # field stores and a 14-tuple cache key per call otherwise, 210 calls per training step on a host that is the bottleneck of the
    # small multi-scale geometries.  The library reads *d during the call only (launch parameters are copied).
    shape = x.shape
    okey = None if out is None else out.shape[-1]
    ckey = (shape, x.dtype, pc.cout_store, pc.w_rows, pc.cin_k, pc.r, pc.s, pc.stem, stride, pad, act, res_mode,
            None if res is None else res.shape, bool(relu_in), float(act_scale), mask is not None, out_hw, okey)
    ent = _CONV_DESCS.get(ckey)
    if ent is None:
        ent = _CONV_DESCS[ckey] = _build_conv_desc(x, pc, stride, pad, act, res, res_mode, relu_in, act_scale, out, out_hw, mask)
    d, oshape, key = ent[0], ent[1], ent[2]
    if out is None:
        out = torch.empty(oshape, device=x.device, dtype=x.dtype)
    if mask is not None:
        assert mask.shape == out.shape and mask.dtype == out.dtype
    args = (_p(x), _p(pc.w), _p(pc.bias), _p(res), _p(mask), _p(act_scale_dev), None, _p(out), _stream())
    if algo is None:
        algo = ALGO_CACHE.get(key)             # (looked up per call: tests and rank > 0 replace the cache's contents)
        if algo is None:
            # (latency-sized shapes are timed with cold weights and the input re-touched: tuner._time_launches_cold)
            algo = _tune(key, d, lambda: _lib.call("osd_conv2d_fwd", C.byref(d), *args), prewarm=lambda: x.float().sum()) if _TUNING[0] else 0
    d.algo = algo
    try:
        _lib.call("osd_conv2d_fwd", C.byref(d), *args)
    except _lib.OsdError as e:
        # a cached id this library does not build for this shape (a cache that slipped past the ABI stamp, a hand-edited file):
        # drop the entry and run the library's own choice instead of failing every step
        if getattr(e, "code", 0) != -2 or algo == 0 or ALGO_CACHE.get(key) != algo:
            raise
        del ALGO_CACHE[key]
        d.algo = 0
        _lib.call("osd_conv2d_fwd", C.byref(d), *args)
    _rec("conv", x=x, w=pc.w, bias=pc.bias, cout=pc.cout_store, r=pc.r, s=pc.s, stem=pc.stem, stride=stride, pad=pad, act=act,
         res=res, res_mode=res_mode, relu_in=bool(relu_in), act_scale=float(act_scale), act_scale_dev=act_scale_dev, mask=mask,
         out=out)
    return out


_CONV_DESCS = {}      # conv2d: (geometry, input shape, epilogue) -> (ConvDesc, output shape, tuner key)


def _p(t):
    """device address of a tensor the CALLER keeps alive across the launch (a named tensor, an attribute): a plain int.  `_ptr` is
    for temporaries built inside an argument list."""
    return None if t is None else t.data_ptr()


def _build_conv_desc(x, pc, stride, pad, act, res, res_mode, relu_in, act_scale, out, out_hw, mask):
    n, h, w, c = x.shape
    d = ConvDesc()
    d.dtype = _dt(x)
    d.n, d.h, d.w = n, h, w
    if pc.stem:
        ho, wo = out_hw
        d.cin, d.r, d.s = 32, 7, 1
        d.in_stride_n, d.in_stride_h, d.in_stride_w = h * w * 4, w * 4, 4
        d.stride_h, d.stride_w, d.pad_h, d.pad_w = 2, 2, 0, 0
        # the 32-element K run of a tap starts at pixel (2*ho + r, 2*wo) and spans 8 padded pixels
        d.h, d.w = h, w - 7
        assert 2 * (wo - 1) + 8 <= w and 2 * (ho - 1) + 7 <= h, "stem input not padded enough"
    else:
        assert c == pc.cin_k, "input channels %d != packed K per tap %d" % (c, pc.cin_k)
        ho, wo = conv_out(h, pc.r, stride, pad), conv_out(w, pc.s, stride, pad)
        d.cin, d.r, d.s = c, pc.r, pc.s
        d.in_stride_n, d.in_stride_h, d.in_stride_w = h * w * c, w * c, c
        d.stride_h = d.stride_w = stride
        d.pad_h = d.pad_w = pad
    d.ho, d.wo, d.cout, d.w_rows = ho, wo, pc.cout_store, pc.w_rows
    oshape = (n, ho, wo, pc.cout_store)
    d.out_stride = pc.cout_store if out is None else out.shape[-1]
    d.res_mode = res_mode
    if res_mode != RES_NONE:
        d.res_h, d.res_w, d.res_stride = res.shape[1], res.shape[2], res.shape[3]
        if res_mode == RES_UP2X:
            assert res.shape[1] * 2 == ho and res.shape[2] * 2 == wo, "top-down map must be exactly half size"
        if res_mode == RES_DOWN2X:
            assert res.shape[1] >= 2 * ho - 1 and res.shape[2] >= 2 * wo - 1, "the every-other-pixel addend must cover the output"
    d.act, d.act_scale, d.relu_in = act, float(act_scale), int(relu_in)
    # full geometry in the tuner key: some algorithms only exist for some map widths (the row-reuse kernel: W in 64/128/256),
    # and a transposed batch (1024x800 after 800x1024) has the same n*ho*wo
    key = (d.dtype, n, ho, wo, d.cout, d.cin, d.r, d.s, d.stride_h, d.pad_h, res_mode, act, int(relu_in), mask is not None)
    return (d, oshape, key)


def _conv2d_two_sources(x, x2, x2_stride, pc, act, out, algo, pc2=None, bias=None):
    """conv3 + downsample of a bottleneck's first block as one GEMM over the concatenated K (osd_conv2d_fwd with src2)."""
    _chk_dev(x, x2)
    n, h, w, c = x.shape
    c2 = x2.shape[-1]
    assert (pc.r, pc.s) == (1, 1) and x2.shape[0] == n and x2.dtype == x.dtype
    if pc2 is None:
        assert c + c2 == pc.cin_k
        bias = pc.bias if bias is None else bias
    else:
        assert (pc.cin_k, pc2.cin_k) == (c, c2) and (pc2.r, pc2.s, pc2.w_rows, pc2.cout_store) == (1, 1, pc.w_rows, pc.cout_store)
        assert bias is not None, "two separately packed convs need their summed bias vector"
    d = ConvDesc()
    d.dtype = _dt(x)
    d.n, d.h, d.w, d.cin, d.r, d.s = n, h, w, c, 1, 1
    d.in_stride_n, d.in_stride_h, d.in_stride_w = h * w * c, w * c, c
    d.stride_h = d.stride_w = 1
    d.pad_h = d.pad_w = 0
    d.ho, d.wo, d.cout, d.w_rows = h, w, pc.cout_store, pc.w_rows
    if out is None:
        out = torch.empty((n, h, w, pc.cout_store), device=x.device, dtype=x.dtype)
    d.out_stride = out.shape[-1]
    d.res_mode = RES_NONE
    d.act, d.act_scale, d.relu_in = act, 1.0, 0
    src2 = ConvSrc2(x2.contiguous().data_ptr(), pc2.w.data_ptr() if pc2 is not None else None, c2, x2.shape[1], x2.shape[2],
                    int(x2_stride))
    args = (_p(x), _p(pc.w), _p(bias), None, None, None, C.byref(src2), _p(out), _stream())
    if algo is None:
        key = ("src2", d.dtype, n, h, w, d.cout, c, c2, int(x2_stride), act, pc2 is not None)
        algo = ALGO_CACHE.get(key)
        if algo is None:
            if _TUNING[0]:
                cands = [1 + v * 8 + t for v in (0, 1, 2, 3) for t in (0, 2, 7)]
                algo = _tune(key, d, lambda: _lib.call("osd_conv2d_fwd", C.byref(d), *args), cands)
            else:
                algo = 0
    d.algo = algo
    _lib.call("osd_conv2d_fwd", C.byref(d), *args)
    _rec("conv", x=x, x2=x2, x2_stride=int(x2_stride), w=pc.w, w2=None if pc2 is None else pc2.w, bias=bias, cout=pc.cout_store,
         r=1, s=1, stem=False, stride=1, pad=0, act=act, res=None, res_mode=RES_NONE, relu_in=False, act_scale=1.0,
         act_scale_dev=None, mask=None, out=out)
    return out




def conv2d_grouped(xs, pc, pad=0, act=ACT_NONE, residuals=None, masks=None, act_scale=1.0, act_scale_devs=None, algo=None,
                   _whole=False):
    """The same stride-1 convolution (shared packed weights pc) over several NHWC tensors of different size — the FPN
    levels of an FCOS tower / prediction conv — in ONE launch (conv2d_multi with one weight pointer for all pairs)."""
    return conv2d_multi(xs, [pc] * len(xs), pad=pad, act=act, residuals=residuals, masks=masks, act_scale=act_scale,
                        act_scale_devs=act_scale_devs, algo=algo, _whole=_whole)


ALGO_SP = 1 + 8 + 6       # tile id 6, variant 1: conv_igemm_sp.hip, the only kernel whose epilogue gathers GroupNorm statistics


def gn_bwd_fusable(x, pc, stride=1, pad=1):
    """Can the data-gradient conv over x (NHWC) gather the GroupNorm-backward statistics of its output pixels
    (osd_conv2d_fwd_multi_gn)?  The software-pipelined 3x3 kernel's shapes, whole 256-pixel tiles, images of whole 128-pixel runs."""
    n, h, w, c = x.shape
    return (x.dtype == torch.bfloat16 and pc.r == 3 and pc.s == 3 and stride == 1 and pad == 1 and w in (64, 128, 256) and c % 64 == 0 and
            pc.cout_store % 256 == 0 and (h * w) % 128 == 0 and (n * h * w) % 256 == 0)


def conv2d_multi(xs, pcs, stride=1, pad=0, act=ACT_NONE, residuals=None, res_mode=None, masks=None, act_scale=1.0,
                 act_scale_devs=None, algo=None, _whole=False, gnb=None):
    """ONE launch of the same conv geometry over several NHWC tensors, each with its own batch / spatial size and its own
    packed weights pcs[i] (osd_conv2d_fwd_multi): the FPN levels of a tower conv (one PackedConv repeated), both towers at
    once, or one layer of the target backbone together with the same layer of the query backbone.  Or TWO launches (large
    segments / small segments, in the given order) where the tuner measured that to be faster: 256-pixel tiles on 256 CUs
    quantise badly (P3..P7 at bs=8 are 534 tiles = 2 full rounds plus a third with 22 tiles).
    residuals: addends (res_mode RES_SAME, default, or RES_UP2X: exactly half size); masks: ReLU-backward masks (data
    gradients); act_scale_devs: one device scalar per segment (the learnable Scale).  Returns the outputs.
    gnb: GroupNorm statistics gathered by the epilogue (osd_conv2d_fwd_multi_gn).  Backward ones (data-gradient convs feeding a
    GroupNorm + ReLU backward): dict(us, abs, gammas, wss, pws: one entry per segment, None where not wanted; n, groups); forward
    ones (the conv whose outputs a GroupNorm normalises): dict(wss; n, groups).  The leading run of segments with an entry goes
    out as its own launch of the software-pipelined kernel, the rest as usual."""
    k = len(xs)
    pc = pcs[0]
    if k == 1 and algo is None and not _whole and gnb is None:      # one pair (every backbone conv of a non-lockstep engine comes through
        # here): straight to the plain launch — all algorithms, its own tuning entry — ahead of the list checks (host time)
        return [conv2d(xs[0], pc, stride=stride, pad=pad, act=act, res=None if residuals is None else residuals[0],
                       res_mode=(RES_NONE if residuals is None else RES_SAME) if res_mode is None else res_mode, act_scale=act_scale,
                       mask=None if masks is None else masks[0],
                       act_scale_dev=None if act_scale_devs is None else act_scale_devs[0])]
    _chk_dev(*xs)
    if gnb is not None and not _whole:
        on = [w is not None for w in gnb["wss"]]
        c = 0
        while c < k and on[c]:
            c += 1
        assert not any(on[c:]), "segments that gather GroupNorm statistics must come first"
        assert residuals is None and masks is None and act == ACT_NONE and act_scale_devs is None
        if c == 0:
            return conv2d_multi(xs, pcs, stride, pad, act, algo=algo)
        head = conv2d_multi(xs[:c], pcs[:c], stride, pad, act, algo=ALGO_SP, _whole=True,
                            gnb={key: (v[:c] if isinstance(v, list) else v) for key, v in gnb.items()})
        return head + (conv2d_multi(xs[c:], pcs[c:], stride, pad, act) if c < k else [])
    assert len(pcs) == k and all((q.cout_store, q.w_rows, q.cin_k, q.r, q.s) == (pc.cout_store, pc.w_rows, pc.cin_k, pc.r, pc.s)
                                 and not q.stem for q in pcs), "segments must share the conv geometry"
    if res_mode is None:
        res_mode = RES_NONE if residuals is None else RES_SAME
    if k == 1 and algo is None and not _whole:       # one pair: the plain launch (all algorithms, its own tuning entry)
        return [conv2d(xs[0], pc, stride=stride, pad=pad, act=act, res=None if residuals is None else residuals[0],
                       res_mode=res_mode, act_scale=act_scale, mask=None if masks is None else masks[0],
                       act_scale_dev=None if act_scale_devs is None else act_scale_devs[0])]
    if k >= 3 and algo is None and not _whole:
        skey = ("split", _dt(xs[0]), tuple(tuple(x.shape) for x in xs), pc.cout_store, pc.r, stride, pad, act, res_mode,
                masks is not None)
        cut = SPLIT_CACHE.get(skey)

        def run(c):
            sl = lambda t, a, b: None if t is None else t[a:b]          # noqa: E731
            if c == 0:
                return conv2d_multi(xs, pcs, stride, pad, act, residuals, res_mode, masks, act_scale, act_scale_devs,
                                    _whole=True)
            return (conv2d_multi(xs[:c], pcs[:c], stride, pad, act, sl(residuals, 0, c), res_mode, sl(masks, 0, c), act_scale,
                                 sl(act_scale_devs, 0, c), _whole=True) +
                    conv2d_multi(xs[c:], pcs[c:], stride, pad, act, sl(residuals, c, k), res_mode, sl(masks, c, k), act_scale,
                                 sl(act_scale_devs, c, k), _whole=True))
        if cut is None and _TUNING[0]:
            timed = []
            for c in range(0, k - 1):
                run(c)                                  # tunes the algorithms of the parts
                timed.append((_time_launches(lambda: run(c)), c))
            SPLIT_CACHE[skey] = cut = _photo_finish(timed, lambda c: _time_launches(lambda: run(c)))
            TUNE_LOG["SPLIT_CACHE"][skey] = sorted(timed)
        if cut is not None:       # (cut 0 = one launch over all segments: also through run(), so that the launch is a `_whole` call)
            return run(cut)
    c = xs[0].shape[-1]
    assert all(x.shape[-1] == pc.cin_k for x in xs), "input channels %d != packed K per tap %d" % (c, pc.cin_k)
    d = ConvDesc()
    d.dtype = _dt(xs[0])
    d.cin, d.r, d.s = c, pc.r, pc.s
    d.stride_h = d.stride_w = stride
    d.pad_h = d.pad_w = pad
    d.cout, d.w_rows = pc.cout_store, pc.w_rows
    outs = [torch.empty((x.shape[0], conv_out(x.shape[1], pc.r, stride, pad), conv_out(x.shape[2], pc.s, stride, pad),
                         pc.cout_store), device=x.device, dtype=x.dtype) for x in xs]
    d.out_stride = pc.cout_store
    d.res_mode = res_mode
    if res_mode != RES_NONE:
        d.res_stride = residuals[0].shape[-1]
        if res_mode == RES_SAME:
            assert all(r.shape == o.shape for r, o in zip(residuals, outs))
        elif res_mode == RES_DOWN2X:
            assert all(r.shape[1] == 2 * o.shape[1] and r.shape[2] == 2 * o.shape[2] and r.shape[3] == o.shape[3]
                       for r, o in zip(residuals, outs)), "the every-other-pixel addend of a grouped launch must be exactly twice the size"
        else:
            assert all(r.shape[1] * 2 == o.shape[1] and r.shape[2] * 2 == o.shape[2] and r.shape[3] == o.shape[3]
                       for r, o in zip(residuals, outs)), "top-down map must be exactly half size"
    if masks is not None:
        assert all(m.shape == o.shape and m.dtype == o.dtype for m, o in zip(masks, outs))
    d.act, d.act_scale, d.relu_in = act, float(act_scale), 0
    ns = (C.c_int32 * k)(*[x.shape[0] for x in xs])
    hs = (C.c_int32 * k)(*[x.shape[1] for x in xs])
    ws = (C.c_int32 * k)(*[x.shape[2] for x in xs])
    args = (k, _ptr_array(xs), _ptr_array(outs), _ptr_array(residuals) if residuals is not None else None,
            _ptr_array(masks) if masks is not None else None,
            _ptr_array(act_scale_devs) if act_scale_devs is not None else None, ns, hs, ws,
            _ptr_array([q.w for q in pcs]), _ptr_array([q.bias for q in pcs]), _stream())

    def launch():
        _lib.call("osd_conv2d_fwd_multi", C.byref(d), *args)
    if gnb is not None:
        d.algo = algo
        bwd = gnb.get("us") is not None        # backward statistics (u, ab, gamma, pw given) or forward ones (sums of the outputs)
        _lib.call("osd_conv2d_fwd_multi_gn", C.byref(d), k, _ptr_array(xs), _ptr_array(outs), ns, hs, ws, _ptr_array([q.w for q in pcs]),
                  _ptr_array([q.bias for q in pcs]), _ptr_array(gnb["us"]) if bwd else None, _ptr_array(gnb["abs"]) if bwd else None,
                  _ptr_array(gnb["gammas"]) if bwd else None, _ptr_array(gnb["wss"]), _ptr_array(gnb["pws"]) if bwd else None,
                  int(gnb["n"]), int(gnb["groups"]), _stream())
        for i in range(k):
            _rec("conv", x=xs[i], w=pcs[i].w, bias=pcs[i].bias, cout=pc.cout_store, r=pc.r, s=pc.s, stem=False, stride=stride, pad=pad,
                 act=act, res=None, res_mode=res_mode, relu_in=False, act_scale=float(act_scale), act_scale_dev=None, mask=None,
                 out=outs[i])
        return outs
    if algo is None:
        key = ("grouped", d.dtype, tuple(tuple(o.shape[:3]) for o in outs), d.cout, d.cin, d.r, d.s, stride, pad, d.res_mode,
               act, masks is not None)
        algo = ALGO_CACHE.get(key)
        if algo is None:
            algo = _tune(key, d, launch) if _TUNING[0] else 0
    d.algo = algo
    launch()
    for i in range(k):
        _rec("conv", x=xs[i], w=pcs[i].w, bias=pcs[i].bias, cout=pc.cout_store, r=pc.r, s=pc.s, stem=False, stride=stride, pad=pad,
             act=act, res=None if residuals is None else residuals[i], res_mode=res_mode, relu_in=False,
             act_scale=float(act_scale), act_scale_dev=None if act_scale_devs is None else act_scale_devs[i],
             mask=None if masks is None else masks[i], out=outs[i])
    return outs


def maxpool3x3s2(x):
    _chk_dev(x)
    n, h, w, c = x.shape
    ho, wo = conv_out(h, 3, 2, 1), conv_out(w, 3, 2, 1)
    y = torch.empty((n, ho, wo, c), device=x.device, dtype=x.dtype)
    _lib.call("osd_maxpool3x3s2_fwd", _p(x), _p(y), n, h, w, c, ho, wo, _dt(x), _stream())
    _rec("maxpool", x=x, out=y)
    return y


def groupnorm_relu(x, gamma, beta, groups=32, eps=1e-5, out=None):
    """relu(GroupNorm(groups, C)(x)) on NHWC x; in place when out is x."""
    _chk_dev(x, gamma, beta)
    n, h, w, c = x.shape
    ws = torch.empty((n, GN_SPLITS, groups, 2), device=x.device, dtype=torch.float32)
    ab = torch.empty((2, n, c), device=x.device, dtype=torch.float32)
    st = _stream()
    _lib.call("osd_groupnorm_stats", _p(x), _p(ws), n, h * w, c, groups, _dt(x), st)
    _lib.call("osd_groupnorm_finalize", _p(ws), _p(gamma), _p(beta), _p(ab[0]), _p(ab[1]), n, h * w, c, groups,
              float(eps), st)
    if out is None:
        out = torch.empty_like(x)
    _lib.call("osd_groupnorm_relu_apply", _p(x), _p(ab[0]), _p(ab[1]), _p(out), n, h * w, c, _dt(x), st)
    return out


def roi_align(x, rois, spatial_scale, ph, pw, sampling_ratio):
    """x NHWC, rois [R,5] fp32 (batch_idx, x1, y1, x2, y2) -> [R, ph, pw, C] fp32."""
    _chk_dev(x, rois)
    n, h, w, c = x.shape
    r = rois.shape[0]
    y = torch.empty((r, ph, pw, c), device=x.device, dtype=torch.float32)
    _lib.call("osd_roialign_fwd", _p(x), _ptr(rois.contiguous().float()), _p(y), n, h, w, c, r, float(spatial_scale),
              ph, pw, sampling_ratio, _dt(x), _stream())
    _rec("roi_align", x=x, rois=rois, scale=float(spatial_scale), ph=ph, pw=pw, sampling_ratio=sampling_ratio, out=y)
    return y


def shot_mean(x, batch):
    """[B*S, C] fp32 -> [B, C] fp32 (batch_pooling)."""
    _chk_dev(x)
    d, c = x.shape
    assert d % batch == 0
    y = torch.empty((batch, c), device=x.device, dtype=torch.float32)
    _lib.call("osd_shot_mean", _ptr(x.contiguous()), _p(y), batch, d // batch, c, _stream())
    _rec("shot_mean", x=x, batch=batch, out=y)
    return y


QUERY_POOL_LEVELS = os.environ.get("OSD_NO_QUERY_POOL_LEVELS", "0") == "0"     # A/B switch: the per-level launches (until round 6)


def query_pool_levels(feats, rois, scales, batch, sampling_ratio):
    """SuppAlignLayer's 1 x 1 ROIAlign of every query's whole-image box + the mean over the shots of a target image
    (generalized_rcnn.py:20-52, 100-104) for ALL FPN levels in one launch: feats[l] NHWC [batch * shots, h, w, C] -> [B, C] fp32 per
    level (views of one buffer).  The arithmetic of roi_align(..., 1, 1, ...) + shot_mean per level."""
    _chk_dev(rois, *feats)
    k = len(feats)
    r, c = feats[0].shape[0], feats[0].shape[-1]
    assert r % batch == 0 and rois.shape[0] == r and all(f.shape[0] == r and f.shape[-1] == c for f in feats)
    flat = torch.empty((k, batch, c), device=feats[0].device, dtype=torch.float32)
    ys = [flat[l] for l in range(k)]
    _lib.call("osd_query_pool_levels", k, _ptr_array(feats), (C.c_int32 * k)(*[f.shape[1] for f in feats]),
              (C.c_int32 * k)(*[f.shape[2] for f in feats]), (C.c_float * k)(*[float(v) for v in scales]), _p(rois), batch, r // batch, c,
              int(sampling_ratio), _ptr_array(ys), _dt(feats[0]), _stream())
    _rec("query_pool", xs=list(feats), rois=rois, scales=[float(v) for v in scales], batch=batch, sampling_ratio=sampling_ratio, outs=ys)
    return ys


def query_pool_levels_bwd(dqs, rois, shapes, scales, shots, sampling_ratio, dtype):
    """Backward of query_pool_levels: dqs[l] [B, C] fp32 -> the gradient of the query feature maps, [B * shots, h, w, C] `dtype` per
    level (views of one buffer): shot_mean_bwd + roi_align_bwd + cast_f32 per level, as three launches for all levels."""
    _chk_dev(rois, *dqs)
    k = len(dqs)
    b, c = dqs[0].shape
    r = b * shots
    dqs = [d.contiguous() for d in dqs]
    sizes = [r * h * w * c for (_, h, w, _) in shapes]
    assert all(tuple(sh) == (r, sh[1], sh[2], c) for sh in shapes) and rois.shape[0] == r
    gx32 = torch.empty((sum(sizes),), device=dqs[0].device, dtype=torch.float32)
    flat = torch.empty((sum(sizes),), device=dqs[0].device, dtype=dtype)
    outs, off = [], 0
    for sh, n in zip(shapes, sizes):
        outs.append(flat[off:off + n].view(tuple(sh)))
        off += n
    _lib.call("osd_query_pool_levels_bwd", k, _ptr_array(dqs), (C.c_int32 * k)(*[sh[1] for sh in shapes]),
              (C.c_int32 * k)(*[sh[2] for sh in shapes]), (C.c_float * k)(*[float(v) for v in scales]), _p(rois), b, shots, c,
              int(sampling_ratio), _p(gx32), _ptr_array(outs), _dt(flat), _stream())
    _rec("query_pool_bwd", dqs=dqs, rois=rois, scales=[float(v) for v in scales], shots=shots, sampling_ratio=sampling_ratio, outs=outs)
    return outs


def correlate(x, q, out=None):
    """x NHWC [N,H,W,C] * q [N,C] fp32 (broadcast over H, W)."""
    _chk_dev(x, q)
    n, h, w, c = x.shape
    assert q.shape == (n, c) and q.dtype == torch.float32
    if out is None:
        out = torch.empty_like(x)
    _lib.call("osd_correlate_fwd", _p(x), _ptr(q.contiguous()), _p(out), n, h * w, c, _dt(x), _stream())
    _rec("correlate", x=x, q=q, out=out)
    return out


def correlate_levels(xs, qs):
    """[x_l * q_l (broadcast over H, W)] for every FPN level in ONE launch (forward correlation, and d_feat = g * q)."""
    _chk_dev(*xs)
    k = len(xs)
    n, c = xs[0].shape[0], xs[0].shape[-1]
    assert all(x.shape[0] == n and x.shape[-1] == c and x.is_contiguous() for x in xs)
    assert all(q.shape == (n, c) and q.dtype == torch.float32 and q.is_contiguous() for q in qs)
    ys = [torch.empty_like(x) for x in xs]
    hws = (C.c_int32 * k)(*[x.shape[1] * x.shape[2] for x in xs])
    _lib.call("osd_correlate_levels", k, _ptr_array(xs), _ptr_array(qs), _ptr_array(ys), hws, n, c, _dt(xs[0]), _stream())
    for x, q, y in zip(xs, qs, ys):
        _rec("correlate", x=x, q=q, out=y)
    return ys


def correlate_bwd_query_levels(gs, feats):
    """dq_l[n][c] = sum_p g_l[n,p,c] * feat_l[n,p,c] for every level in one launch -> list of [n, c] fp32."""
    k = len(gs)
    n, c = gs[0].shape[0], gs[0].shape[-1]
    dq = torch.empty((k, n, c), device=gs[0].device, dtype=torch.float32)
    dqs = [dq[l] for l in range(k)]
    hws = (C.c_int32 * k)(*[g.shape[1] * g.shape[2] for g in gs])
    _lib.call("osd_correlate_bwd_query_levels", k, _ptr_array(gs), _ptr_array(feats), _ptr_array(dqs), hws, n, c, _dt(gs[0]),
              _stream())
    for g, f, o in zip(gs, feats, dqs):
        _rec("correlate_bwd_query", g=g, feat=f, out=o)
    return dqs


def nhwc_to_nchw_f32(x, c0=0, c=None):
    _chk_dev(x)
    n, h, w, stride = x.shape
    c = stride - c0 if c is None else c
    y = torch.empty((n, c, h, w), device=x.device, dtype=torch.float32)
    _lib.call("osd_nhwc_to_nchw_f32", _p(x), _p(y), n, h, w, c, stride, c0, _dt(x), _stream())
    return y


def nchw_f32_to_nhwc(x, dtype):
    _chk_dev(x)
    n, c, h, w = x.shape
    y = torch.empty((n, h, w, c), device=x.device, dtype=dtype)
    _lib.call("osd_nchw_f32_to_nhwc", _ptr(x.contiguous().float()), _p(y), n, c, h, w, _dt(y), _stream())
    return y


def fcos_score_decode(cls_ctr, reg, scores, boxes, stride, loc_offset, img_h, img_w, img_hw=None):
    """img_hw: optional [N,2] fp32 device tensor of true (height, width) per image (padded batches)."""
    _chk_dev(cls_ctr, reg, scores, boxes, img_hw)
    n, h, w, ccs = cls_ctr.shape
    _lib.call("osd_fcos_score_decode_sizes", _p(cls_ctr), _p(reg), _p(scores), _p(boxes), n, h, w, ccs,
              reg.shape[-1], stride, loc_offset, scores.shape[1], float(img_h), float(img_w), _p(img_hw), _dt(cls_ctr),
              _stream())


def level_topk(keys, lo, cnt, topn):
    n, total = keys.shape
    _lib.call("osd_level_topk", _p(keys), _p(keys), n, total, lo, cnt, topn, _stream())


def rank_sort_gather(keys, boxes, max_count, levels=None, topn=0):
    """keys [N,T] fp32 (dropped = -1), boxes [N,T,4]; levels = [(lo, cnt), ...] tiling [0,T) with a per-level top-`topn`
    cut (None: no cut) -> boxes_sorted [N,max_count,4], scores_sorted, idx_sorted, counts."""
    _chk_dev(keys, boxes)
    n, total = keys.shape
    dev = keys.device
    bs = torch.empty((n, max_count, 4), device=dev, dtype=torch.float32)
    ss = torch.empty((n, max_count), device=dev, dtype=torch.float32)
    idx = torch.empty((n, max_count), device=dev, dtype=torch.int32)
    cnt = torch.empty((n,), device=dev, dtype=torch.int32)
    if levels:
        lo = (C.c_int32 * len(levels))(*[l for l, _ in levels])
        lc = (C.c_int32 * len(levels))(*[c for _, c in levels])
        nl = len(levels)
    else:
        lo, lc, nl = None, None, 0
    _lib.call("osd_rank_sort_gather", _p(keys), _p(boxes), n, total, max_count, lo, lc, nl, int(topn), _p(bs),
              _p(ss), _p(idx), _p(cnt), _stream())
    return bs, ss, idx, cnt


def nms_sorted(boxes_sorted, scores_sorted, counts, thresh, max_keep, cuda_semantics=False, workspace=None):
    _chk_dev(boxes_sorted, scores_sorted, counts)
    n, max_count, _ = boxes_sorted.shape
    dev = boxes_sorted.device
    need = _lib.load().osd_nms_workspace_bytes(n, max_count)
    if workspace is None or workspace.numel() * workspace.element_size() < need:
        workspace = torch.empty((max(need, 8) // 8,), device=dev, dtype=torch.int64)
    ob = torch.zeros((n, max_keep, 4), device=dev, dtype=torch.float32)
    os_ = torch.zeros((n, max_keep), device=dev, dtype=torch.float32)
    op = torch.zeros((n, max_keep), device=dev, dtype=torch.int32)
    oc = torch.empty((n,), device=dev, dtype=torch.int32)
    _lib.call("osd_nms_sorted", _p(boxes_sorted), _p(scores_sorted), _p(counts), n, max_count, float(thresh),
              int(cuda_semantics), max_keep, _p(workspace), _p(ob), _p(os_), _p(op), _p(oc), _stream())
    return ob, os_, op, oc


def nms(dets, scores, thresh, cuda_semantics=True):
    """_C.nms in one call: dets [N,4] fp32, scores [N] fp32 -> (keep [N] int64 ascending original indices, count [1] int32
    on the device; keep[:count] is the result)."""
    _chk_dev(dets, scores)
    n = dets.shape[0]
    dev = dets.device
    keep = torch.empty((n,), device=dev, dtype=torch.int64)
    count = torch.empty((1,), device=dev, dtype=torch.int32)
    need = int(_lib.load().osd_nms_single_workspace_bytes(n))
    ws = torch.empty((need // 8 + 1,), device=dev, dtype=torch.int64)
    _lib.call("osd_nms", _p(dets), _p(scores), n, float(thresh), int(cuda_semantics), _p(ws), _p(keep), _p(count),
              _stream())
    return keep, count


def sigmoid_focal_loss_fwd(logits, targets, gamma, alpha):
    _chk_dev(logits, targets)
    m, classes = logits.shape
    losses = torch.empty_like(logits)
    _lib.call("osd_sigmoid_focal_fwd", _ptr(logits.contiguous()), _ptr(targets.contiguous()), _p(losses), m, classes,
              float(gamma), float(alpha), _stream())
    return losses


def sigmoid_focal_loss_bwd(logits, targets, d_losses, gamma, alpha):
    m, classes = logits.shape
    d_logits = torch.empty_like(logits)
    _lib.call("osd_sigmoid_focal_bwd", _ptr(logits.contiguous()), _ptr(targets.contiguous()),
              _ptr(d_losses.contiguous()), _p(d_logits), m, classes, float(gamma), float(alpha), _stream())
    return d_logits


# ---------------------------------------------------------------------------------------------- backward wrappers
def _conv_desc(x_shape, dtype_code, cout_store, r, s, stride, pad, out_stride):
    n, h, w, c = x_shape
    d = ConvDesc()
    d.dtype = dtype_code
    d.n, d.h, d.w, d.cin = n, h, w, c
    d.in_stride_n, d.in_stride_h, d.in_stride_w = h * w * c, w * c, c
    d.ho, d.wo = conv_out(h, r, stride, pad), conv_out(w, s, stride, pad)
    d.cout, d.r, d.s = cout_store, r, s
    d.stride_h = d.stride_w = stride
    d.pad_h = d.pad_w = pad
    d.out_stride = out_stride
    return _wgrad_ws(d)            # (only the weight-gradient entries read these fields)


def pack_conv_master(w_orsi, scale, dtype, w_rows=None):
    """fp32 master weight in [cout][r][s][cin] order (+ optional FrozenBN scale) -> forward PackedConv layout."""
    cout, r, s, cin = w_orsi.shape
    w_rows = _round_up(cout, 16) if w_rows is None else w_rows
    mult = 64 if dtype == torch.bfloat16 else 16
    cin_pad = _round_up(cin, mult)
    wp = torch.empty((w_rows, r, s, cin_pad), device=w_orsi.device, dtype=dtype)
    _lib.call("osd_pack_conv_weight_ex", _p(w_orsi), _p(scale), _p(wp), cout, cin, r, s, w_rows, cin_pad, 1, _dt(wp),
              _stream())
    return wp


def pack_conv_master_dgrad(w_orsi, scale, dtype):
    """-> [cin_pad16][r][s][cout_pad] (taps flipped): the weights of the data-gradient convolution."""
    cout, r, s, cin = w_orsi.shape
    rows = _round_up(cin, 16)
    mult = 64 if dtype == torch.bfloat16 else 16
    cout_pad = _round_up(cout, mult)
    wp = torch.empty((rows, r, s, cout_pad), device=w_orsi.device, dtype=dtype)
    _lib.call("osd_pack_conv_weight_dgrad", _p(w_orsi), _p(scale), _p(wp), cout, cin, r, s, rows, cout_pad, 1,
              _dt(wp), _stream())
    return wp




_WGRAD_WS = {}      # stream handle -> the tensor registered as its ordered-mode workspace (kept alive here)


def wgrad_set_workspace(stream=None, nbytes=1 << 30):
    """Ordered mode of the weight-gradient launches made on `stream` (default: the current one): partial tiles as plain stores
    into a scratch buffer + a fixed-order reduction pass = bit-reproducible dW.  The buffer is kept HERE, on the host side, and
    handed to the library with every call (osd_conv_desc.ordered_ws): the library has no per-stream state.  nbytes = 0 switches
    back to atomics.  Tuner results are cached per mode."""
    st = stream if stream is not None else torch.cuda.current_stream()
    if nbytes:
        _WGRAD_WS[st.cuda_stream] = torch.empty((nbytes // 4,), device=st.device, dtype=torch.float32)
    else:
        _WGRAD_WS.pop(st.cuda_stream, None)


def _wgrad_ws(d):
    """Fill the descriptor's ordered-mode fields from the current stream's registered scratch buffer (if any)."""
    buf = _WGRAD_WS.get(_stream_handle())
    if buf is not None:
        d.ordered_ws, d.ordered_ws_bytes = buf.data_ptr(), buf.numel() * 4
    return d


def _wgrad_mode():
    """Part of the weight-gradient tuner's cache key: the best variant differs between atomics and ordered mode."""
    return _stream_handle() in _WGRAD_WS


def conv2d_wgrad(x, dy, dw_packed, r, s, stride, pad, cout, scale=None, db=None, algo=None):
    """dw_packed [cout][r][s][cin] fp32 += wgrad(x NHWC, dy NHWC [N,Ho,Wo,>=cout]); db [cout] fp32 += sum_m dy (optional)."""
    _chk_dev(x, dy, dw_packed)
    d = _conv_desc(x.shape, _dt(x), cout, r, s, stride, pad, dy.shape[-1])
    assert (dy.shape[1], dy.shape[2]) == (d.ho, d.wo), (dy.shape, d.ho, d.wo)
    st = _stream()

    def launch(dw, dbias):
        _lib.call("osd_conv2d_wgrad", C.byref(d), _p(x), _p(dy), _p(scale), _p(dw), _p(dbias), st)
    key = (d.dtype, tuple(x.shape), cout, r, s, stride, pad, dy.shape[-1], db is not None, _wgrad_mode())
    if algo is None:
        algo = WGRAD_ALGO_CACHE.get(key)
    if algo is None:
        algo = _tune_wgrad(key, d, launch, dw_packed, db, [x.shape[2]]) if _TUNING[0] else 0
    d.algo = algo
    launch(dw_packed, db)
    _rec("wgrad", items=[dict(x=x, dy=dy, dw=dw_packed, scale=scale, db=db, r=r, s=s, stride=stride, pad=pad, cout=cout)])


PRED_G = 64      # columns of the prediction convs' gathered dy matrix (osd_pred_dy_gather)


def pred_gemm_ok(cout, r, s, stride, pad, cin, n_levels):
    """Do the prediction convs' GEMM paths (weight gradient and data gradient over the gathered dy matrix G) cover this conv?"""
    return cout <= 4 and (r, s, stride, pad) == (3, 3, 1, 1) and cin % 256 == 0 and n_levels <= 6


def pred_dy_gather(dys, cout, cin):
    """G[q][tap * 4 + co] = dy[q - (tap / 3 - 1, tap % 3 - 1)][co] for every pixel q of the levels `dys` (NHWC, >= 4 channels stored), the
    levels one after the other: [sum n h w][64].  Shared by the prediction convs' weight gradient and data gradient."""
    dy0 = dys[0]
    d = _conv_desc((dy0.shape[0], dy0.shape[1], dy0.shape[2], cin), _dt(dy0), cout, 3, 3, 1, 1, dy0.shape[-1])
    k = len(dys)
    tot = sum(t.shape[0] * t.shape[1] * t.shape[2] for t in dys)
    g = torch.empty((tot, PRED_G), device=dy0.device, dtype=dy0.dtype)
    ptrs = (C.c_void_p * k)(*[t.data_ptr() for t in dys])
    ns = (C.c_int32 * k)(*[t.shape[0] for t in dys])
    hs = (C.c_int32 * k)(*[t.shape[1] for t in dys])
    ws = (C.c_int32 * k)(*[t.shape[2] for t in dys])
    _lib.call("osd_pred_dy_gather", C.byref(d), k, ptrs, ns, hs, ws, _p(g), _stream())
    _rec("pred_gather", dys=list(dys), cout=cout, out=g)
    return g


def pred_dgrad_pack(w_master, cout, cin, dtype, out=None):
    """The prediction conv's fp32 master [cout][3][3][cin] as the packed weights of the 1x1 data-gradient conv over G: a PackedConv
    with cin rows of 64 columns (column tap * 4 + co)."""
    wd = torch.empty((cin, 1, 1, PRED_G), device=w_master.device, dtype=dtype) if out is None else out
    _lib.call("osd_pred_dgrad_pack", _dt(wd), _p(w_master), cout, cin, _p(wd), _stream())
    _rec("pred_dgrad_pack", w=w_master, cout=cout, cin=cin, out=wd)
    return wd


def conv2d_wgrad_grouped(pairs, dw_packed, r, s, stride, pad, cout, scale=None, db=None, algo=None, g=None):
    """One launch over several (x, dy) pairs sharing the weights (FPN levels): dw_packed += sum over pairs.
    g: the prediction convs' gathered dy matrix (pred_dy_gather) when the caller has built it already (the data gradient shares it)."""
    x0, dy0 = pairs[0]
    d = _conv_desc(x0.shape, _dt(x0), cout, r, s, stride, pad, dy0.shape[-1])
    k = len(pairs)
    xs = (C.c_void_p * k)(*[x.data_ptr() for x, _ in pairs])
    dys = (C.c_void_p * k)(*[dy.data_ptr() for _, dy in pairs])
    ns = (C.c_int32 * k)(*[x.shape[0] for x, _ in pairs])
    hs = (C.c_int32 * k)(*[x.shape[1] for x, _ in pairs])
    ws = (C.c_int32 * k)(*[x.shape[2] for x, _ in pairs])
    st = _stream()
    if (cout <= 4 and (r, s, stride, pad) == (3, 3, 1, 1) and x0.shape[-1] % 256 == 0 and scale is None and k <= 6
            and not os.environ.get("OSD_NO_PRED_WGRAD")):
        # prediction convs (2 / 4 output channels): the read-once kernel, not a 128-channel MFMA tile
        if g is not None and not os.environ.get("OSD_PRED_WGRAD_READONCE"):
            wsp = torch.empty((64 + PRED_G * x0.shape[-1] + 64,), device=x0.device, dtype=torch.float32)
            _lib.call("osd_conv2d_wgrad_pred_gathered", C.byref(d), k, xs, _p(g), ns, hs, ws, _p(dw_packed), _p(db), _p(wsp), st)
        else:
            need = int(_lib.load().osd_conv2d_wgrad_pred_workspace_bytes(k, ns, hs, ws, x0.shape[-1]))
            wsp = torch.empty((need // 4 + 1,), device=x0.device, dtype=torch.float32)
            _lib.call("osd_conv2d_wgrad_pred", C.byref(d), k, xs, dys, ns, hs, ws, _p(dw_packed), _p(db), _p(wsp), st)
        _rec("wgrad", items=[dict(x=x, dy=dy, dw=dw_packed, scale=None, db=db, r=r, s=s, stride=stride, pad=pad, cout=cout)
                             for x, dy in pairs])
        return

    def launch(dw, dbias):
        _lib.call("osd_conv2d_wgrad_grouped", C.byref(d), k, xs, dys, ns, hs, ws, _p(scale), _p(dw), _p(dbias), st)
    key = (d.dtype, tuple(tuple(x.shape) for x, _ in pairs), cout, r, s, stride, pad, dy0.shape[-1], db is not None, _wgrad_mode())
    if algo is None:
        algo = WGRAD_ALGO_CACHE.get(key)
    if algo is None:
        algo = _tune_wgrad(key, d, launch, dw_packed, db, [x.shape[2] for x, _ in pairs]) if _TUNING[0] else 0
    d.algo = algo
    launch(dw_packed, db)
    _rec("wgrad", items=[dict(x=x, dy=dy, dw=dw_packed, scale=scale, db=db, r=r, s=s, stride=stride, pad=pad, cout=cout)
                         for x, dy in pairs])


def conv2d_wgrad_batched(items, r, s, stride, pad, cout, algo=None):
    """items: [(x, dy, dw, scale or None, db or None)] — convs of identical geometry with their own weights (the repeated
    bottleneck blocks of a stage): ONE launch, dw_i += wgrad(x_i, dy_i)."""
    x0, dy0 = items[0][0], items[0][1]
    assert all(it[0].shape == x0.shape and it[1].shape == dy0.shape for it in items)
    d = _conv_desc(x0.shape, _dt(x0), cout, r, s, stride, pad, dy0.shape[-1])
    k = len(items)
    xs = (C.c_void_p * k)(*[it[0].data_ptr() for it in items])
    dys = (C.c_void_p * k)(*[it[1].data_ptr() for it in items])
    scales = (C.c_void_p * k)(*[(it[3].data_ptr() if it[3] is not None else 0) for it in items])
    st = _stream()
    has_db = any(it[4] is not None for it in items)

    def launch(dws, dbs):
        dwp = (C.c_void_p * k)(*[t.data_ptr() for t in dws])
        dbp = (C.c_void_p * k)(*[(t.data_ptr() if t is not None else 0) for t in dbs])
        _lib.call("osd_conv2d_wgrad_batched", C.byref(d), k, xs, dys, scales, dwp, dbp, st)
    key = ("batched", k, d.dtype, tuple(x0.shape), cout, r, s, stride, pad, dy0.shape[-1], has_db, _wgrad_mode())
    if algo is None:
        algo = WGRAD_ALGO_CACHE.get(key)
    if algo is None:
        if _TUNING[0]:
            sdw = [torch.empty_like(it[2]) for it in items]
            sdb = [None if it[4] is None else torch.empty_like(it[4]) for it in items]
            algo = _tune_wgrad(key, d, lambda a, b: launch(sdw, sdb), items[0][2], None, [x0.shape[2]])
        else:
            algo = 0
    d.algo = algo
    launch([it[2] for it in items], [it[4] for it in items])
    _rec("wgrad", items=[dict(x=it[0], dy=it[1], dw=it[2], scale=it[3], db=it[4], r=r, s=s, stride=stride, pad=pad, cout=cout)
                         for it in items])


def conv2d_wgrad_multi(items, r, s, stride, pad, cout, algo=None):
    """items: [(x, dy, dw, scale or None, db or None)] with the same conv geometry (channels, kernel, stride, pad) but any
    batch / spatial size per item; items may share a dw (FPN levels of one conv) or not (different convs): ONE launch."""
    x0, dy0 = items[0][0], items[0][1]
    d = _conv_desc(x0.shape, _dt(x0), cout, r, s, stride, pad, dy0.shape[-1])
    k = len(items)
    xs = (C.c_void_p * k)(*[it[0].data_ptr() for it in items])
    dys = (C.c_void_p * k)(*[it[1].data_ptr() for it in items])
    ns = (C.c_int32 * k)(*[it[0].shape[0] for it in items])
    hs = (C.c_int32 * k)(*[it[0].shape[1] for it in items])
    ws = (C.c_int32 * k)(*[it[0].shape[2] for it in items])
    scales = (C.c_void_p * k)(*[(it[3].data_ptr() if it[3] is not None else 0) for it in items])
    st = _stream()
    real_dws, real_dbs = [it[2] for it in items], [it[4] for it in items]

    def launch(dws, dbs):
        dwp = (C.c_void_p * k)(*[t.data_ptr() for t in dws])
        dbp = (C.c_void_p * k)(*[(t.data_ptr() if t is not None else 0) for t in dbs])
        _lib.call("osd_conv2d_wgrad_multi", C.byref(d), k, xs, dys, ns, hs, ws, scales, dwp, dbp, st)
    key = ("multi", d.dtype, tuple(tuple(it[0].shape) for it in items), tuple(id(it[2]) == id(items[0][2]) for it in items),
           cout, r, s, stride, pad, dy0.shape[-1], any(b is not None for b in real_dbs), _wgrad_mode())
    if algo is None:
        algo = WGRAD_ALGO_CACHE.get(key)
    if algo is None:
        if _TUNING[0]:
            scratch = {}
            sdw = [scratch.setdefault(id(t), torch.empty_like(t)) for t in real_dws]
            sdb = [None if t is None else scratch.setdefault(id(t), torch.empty_like(t)) for t in real_dbs]
            algo = _tune_wgrad(key, d, lambda a, b: launch(sdw, sdb), real_dws[0], None, [it[0].shape[2] for it in items])
        else:
            algo = 0
    d.algo = algo
    launch(real_dws, real_dbs)
    _rec("wgrad", items=[dict(x=it[0], dy=it[1], dw=it[2], scale=it[3], db=it[4], r=r, s=s, stride=stride, pad=pad, cout=cout)
                         for it in items])


def conv2d_wgrad_mixed(items, algo=None):
    """items: [(x, dy, dw, scale or None, db or None, r, s, stride, pad, cout)] — convs of ANY geometry (the weight
    gradients of a whole ResNet stage): ONE launch, the workgroups shared out in proportion to pixels x output tiles."""
    k = len(items)
    descs = (ConvDesc * k)()
    for i, (x, dy, dw, scale, db, r, s, stride, pad, cout) in enumerate(items):
        d = _conv_desc(x.shape, _dt(x), cout, r, s, stride, pad, dy.shape[-1])
        assert (dy.shape[1], dy.shape[2]) == (d.ho, d.wo), (x.shape, dy.shape)
        descs[i] = d
    xs = (C.c_void_p * k)(*[it[0].data_ptr() for it in items])
    dys = (C.c_void_p * k)(*[it[1].data_ptr() for it in items])
    scales = (C.c_void_p * k)(*[(it[3].data_ptr() if it[3] is not None else 0) for it in items])
    st = _stream()
    real_dws, real_dbs = [it[2] for it in items], [it[4] for it in items]

    def launch(dws, dbs):
        dwp = (C.c_void_p * k)(*[t.data_ptr() for t in dws])
        dbp = (C.c_void_p * k)(*[(t.data_ptr() if t is not None else 0) for t in dbs])
        _lib.call("osd_conv2d_wgrad_mixed", k, descs, xs, dys, scales, dwp, dbp, st)
    key = ("mixed", descs[0].dtype, tuple((tuple(it[0].shape), tuple(it[1].shape)) + tuple(it[5:]) for it in items),
           tuple([id(u) for u in real_dws].index(id(t)) for t in real_dws), _wgrad_mode())
    if algo is None:
        algo = WGRAD_ALGO_CACHE.get(key)
    if algo is None:
        if _TUNING[0]:
            scratch = {}
            sdw = [scratch.setdefault(id(t), torch.empty_like(t)) for t in real_dws]
            sdb = [None if t is None else scratch.setdefault(id(t), torch.empty_like(t)) for t in real_dbs]
            timed = []
            timer = wgrad_timer()
            for cand in wgrad_algo_candidates(descs[0].dtype, max(d.cout for d in descs), max(d.cin for d in descs)):
                descs[0].algo = cand
                if not _candidate_runs(lambda: launch(sdw, sdb)):
                    continue
                timed.append((timer(lambda: launch(sdw, sdb)), cand))

            def retime(cand):
                descs[0].algo = cand
                return timer(lambda: launch(sdw, sdb))
            WGRAD_ALGO_CACHE[key] = algo = _photo_finish(timed, retime)
            TUNE_LOG["WGRAD_ALGO_CACHE"][key] = sorted(timed)
        else:
            algo = 0
    descs[0].algo = algo
    launch(real_dws, real_dbs)
    _rec("wgrad", items=[dict(x=it[0], dy=it[1], dw=it[2], scale=it[3], db=it[4], r=it[5], s=it[6], stride=it[7], pad=it[8],
                              cout=it[9]) for it in items])


def bias_grad(dy, db, c):
    n, h, w, stride = dy.shape
    _lib.call("osd_bias_grad", _p(dy), _p(db), n * h * w, c, stride, _dt(dy), _stream())


def conv2d_dgrad_naive(dy, w_fwd_packed, x_shape, r, s, stride, pad, cout, mask=None, addend=None):
    d = _conv_desc(x_shape, _dt(dy), cout, r, s, stride, pad, dy.shape[-1])
    dx = torch.empty(x_shape, device=dy.device, dtype=dy.dtype)
    _lib.call("osd_conv2d_dgrad_naive", C.byref(d), _p(dy), _p(w_fwd_packed), _p(mask), _p(addend), _p(dx),
              _stream())
    return dx


def scatter2x(src, out_hw, mask=None, addend=None):
    n, ho, wo, c = src.shape
    h, w = out_hw
    dst = torch.empty((n, h, w, c), device=src.device, dtype=src.dtype)
    _lib.call("osd_scatter2x", _p(src), _p(mask), _p(addend), _p(dst), n, h, w, ho, wo, c, _dt(src), _stream())
    _rec("scatter2x", x=src, mask=mask, addend=addend, out=dst)
    return dst


def add_mask(a, b=None, mask=None, out=None):
    out = torch.empty_like(a) if out is None else out
    _lib.call("osd_add_mask", _p(a), _p(b), _p(mask), _p(out), a.numel(), _dt(a), _stream())
    _rec("add_mask", a=a, b=b, mask=mask, out=out)
    return out


def upsample2x_bwd(inner, prev=None):
    n, h2, w2, c = inner.shape
    top = torch.empty((n, h2 // 2, w2 // 2, c), device=inner.device, dtype=inner.dtype)
    _lib.call("osd_upsample2x_bwd", _p(inner), _p(prev), _p(top), n, h2 // 2, w2 // 2, c, _dt(inner), _stream())
    _rec("upsample2x_bwd", inner=inner, prev=prev, out=top)
    return top


def correlate_bwd_query(g, feat):
    n, h, w, c = g.shape
    dq = torch.empty((n, c), device=g.device, dtype=torch.float32)
    _lib.call("osd_correlate_bwd_query", _p(g), _p(feat), _p(dq), n, h * w, c, _dt(g), _stream())
    _rec("correlate_bwd_query", g=g, feat=feat, out=dq)
    return dq


def roi_align_bwd(gy, rois, x_shape, spatial_scale, ph, pw, sampling_ratio):
    b, h, w, c = x_shape
    gx = torch.empty(x_shape, device=gy.device, dtype=torch.float32)
    _lib.call("osd_roialign_bwd", _ptr(gy.contiguous()), _p(rois), _p(gx), b, h, w, c, rois.shape[0],
              float(spatial_scale), ph, pw, sampling_ratio, _stream())
    _rec("roi_align_bwd", gy=gy, rois=rois, scale=float(spatial_scale), ph=ph, pw=pw, sampling_ratio=sampling_ratio, out=gx)
    return gx


def shot_mean_bwd(gy, shots):
    b, c = gy.shape
    gx = torch.empty((b * shots, c), device=gy.device, dtype=torch.float32)
    _lib.call("osd_shot_mean_bwd", _ptr(gy.contiguous()), _p(gx), b, shots, c, _stream())
    _rec("shot_mean_bwd", gy=gy, shots=shots, out=gx)
    return gx


def cast_f32(x, dtype):
    y = torch.empty(x.shape, device=x.device, dtype=dtype)
    _lib.call("osd_cast_f32", _ptr(x.contiguous()), _p(y), x.numel(), _dt(y), _stream())
    _rec("cast", x=x, out=y)
    return y


def groupnorm_relu_train(x, gamma, beta, groups=32, eps=1e-5):
    """relu(GroupNorm(x)) of ONE tensor keeping what the backward pass needs: the one-level case of groupnorm_relu_levels."""
    ys, ab = groupnorm_relu_levels([x], gamma, beta, groups, eps)
    return ys[0], ab


def groupnorm_relu_bwd(u, dt, ab, gamma, beta, dgamma, dbeta, groups=32):
    return groupnorm_relu_bwd_levels([u], [dt], ab, gamma, beta, dgamma, dbeta, groups)[0]


def fcos_loss_level(phase, cls_ctr, reg, gt_boxes, gt_count, stride, size_lo, size_hi, radius, gamma, alpha, scale_dev,
                    sums, d_cls_ctr=None, d_reg=None, d_scale_raw=None):
    n, h, w, _ = cls_ctr.shape
    gs = d_cls_ctr.shape[-1] if d_cls_ctr is not None else 4
    _lib.call("osd_fcos_loss_level", phase, _p(cls_ctr), _p(reg), _p(gt_boxes), _p(gt_count), gt_boxes.shape[1], n,
              h, w, stride, float(size_lo), float(size_hi), float(radius), float(gamma), float(alpha), _p(scale_dev),
              _p(sums), _p(d_cls_ctr), _p(d_reg), gs, _p(d_scale_raw), _dt(cls_ctr), _stream())


def _ptr_array(tensors):
    return (C.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


def fcos_loss_levels(phase, head_out, gt_boxes, gt_count, strides, size_ranges, radius, gamma, alpha, scale_devs, sums,
                     d_cls_ctrs=None, d_regs=None, d_scale_raws=None):
    """fcos_loss_level for every FPN level in ONE launch.  head_out: [(cls_ctr, reg)] per level."""
    k = len(head_out)
    n = head_out[0][0].shape[0]
    gs = d_cls_ctrs[0].shape[-1] if d_cls_ctrs is not None else 4
    hs = (C.c_int32 * k)(*[c.shape[1] for c, _ in head_out])
    ws = (C.c_int32 * k)(*[c.shape[2] for c, _ in head_out])
    st = (C.c_int32 * k)(*strides)
    lo = (C.c_float * k)(*[float(a) for a, _ in size_ranges])
    hi = (C.c_float * k)(*[float(b) for _, b in size_ranges])
    none = C.c_void_p(0)
    _lib.call("osd_fcos_loss_levels", phase, k, _ptr_array([c for c, _ in head_out]), _ptr_array([r for _, r in head_out]),
              _p(gt_boxes), _p(gt_count), gt_boxes.shape[1], n, hs, ws, st, lo, hi, float(radius), float(gamma), float(alpha),
              _ptr_array(scale_devs) if scale_devs is not None else none, _p(sums),
              _ptr_array(d_cls_ctrs) if d_cls_ctrs is not None else none, _ptr_array(d_regs) if d_regs is not None else none,
              gs, _ptr_array(d_scale_raws) if d_scale_raws is not None else none, _dt(head_out[0][0]), _stream())
    _rec("fcos_loss", phase=phase, head_out=list(head_out), gt_boxes=gt_boxes, gt_count=gt_count, gamma=float(gamma), alpha=float(alpha),
         scale_devs=scale_devs, sums=sums, d_cls_ctrs=d_cls_ctrs, d_regs=d_regs, d_scale_raws=d_scale_raws)


def gn_fwd_ws_parts(ws, k, n, groups=32):
    """-> per level views of the forward ws [k][n][GN_SPLITS][groups][2]: what osd_conv2d_fwd_multi_gn (forward) accumulates into"""
    a = n * GN_SPLITS * groups * 2
    return [ws[l * a:(l + 1) * a] for l in range(k)]


# ---- one-pass GroupNorm (csrc/groupnorm_onepass.hip): OSD_GN_ONEPASS=0 keeps the two-launch kernels ----
# OSD_GN_ONEPASS: "f" (default) = the forward kernel only — +0.8 % on the bs = 8 step in same-box A/B, the backward kernel -1 % .. +0.3 %
# (profiles/r5_gn1p_ab.txt: its 1,160 workgroups are 2.3 rounds of load / hand-off / store, and the two-launch form's second read
# of (u, dt) comes out of the Infinity Cache, not HBM); "1" = both, "b" = backward only, "0" = the two-launch kernels
_GN1P_MODE = os.environ.get("OSD_GN_ONEPASS", "f")
GN_ONEPASS = _GN1P_MODE != "0"
GN_ONEPASS_FWD = _GN1P_MODE in ("f", "1")
GN_ONEPASS_BWD = _GN1P_MODE in ("b", "1")
_GN1P_SYNC = {}      # (device index, stream handle) -> zeroed int32 words: launches on one stream are ordered, streams never share


def gn_onepass_ok(x, groups):
    c = x.shape[-1]
    return (GN_ONEPASS and x.dtype == torch.bfloat16 and c % 8 == 0 and c <= 512 and 512 % (c // 8) == 0 and c % groups == 0
            and (c // groups) % 8 == 0 and groups <= 64)


def _gn1p_sync(dev, k, n):
    need = int(_lib.load().osd_groupnorm_onepass_sync_bytes(k, n)) // 4
    # keyed on the CURRENT stream, which is also the stream _stream() hands to the launch that follows (both read
    # torch.cuda.current_stream() inside one `with torch.cuda.stream(...)` scope: no stream switch between the two calls)
    key = (dev.index, _stream_handle())
    buf = _GN1P_SYNC.get(key)
    if buf is None or buf.numel() < need:
        buf = _GN1P_SYNC[key] = torch.zeros((max(need, 32 * 513),), device=dev, dtype=torch.int32)
    return buf


def gn_onepass_errors():
    """sum of the error words of every one-pass sync buffer (0 = no workgroup ever gave up waiting); synchronises"""
    return int(sum(int(b[2].item()) for b in _GN1P_SYNC.values()))


def gn_onepass_check(where=""):
    """raise when any one-pass GroupNorm launch timed out waiting for a workgroup of its job (such a launch has also written NaN
    into its outputs: groupnorm_onepass.hip).  Synchronises: call it where the host synchronises anyway (bench.py after the timed
    region, TrainEngine.state_dict, the end of an evaluation loop), never per step."""
    if _GN1P_SYNC and gn_onepass_errors() != 0:
        raise _lib.OsdError("one-pass GroupNorm%s: a workgroup timed out waiting for its (level, image)'s partial sums — the outputs "
                            "of that launch are NaN; set OSD_GN_ONEPASS=0 for the two-launch kernels" % (" (%s)" % where if where else ""))


def _gn1p_ws(dev, k, hws, n, c, groups, backward):
    nbytes = int(_lib.load().osd_groupnorm_onepass_workspace_bytes(k, hws, n, c, groups, int(backward)))
    assert nbytes > 0
    return torch.empty((nbytes // 4,), device=dev, dtype=torch.float32)


def groupnorm_relu_levels(xs, gamma, beta, groups=32, eps=1e-5, ws=None, fused_mask=0):
    """relu(GroupNorm(x_l)) for the FPN levels of one tower layer in two launches.  Returns (ys, ab) with
    ab [L][4][N][C] fp32 (per-image scale/shift a, b and the normalisation xa, xb, kept for the backward pass).
    ws / fused_mask: levels whose sums the conv that wrote xs has already added to the (zeroed) ws skip the statistics pass."""
    n, _, _, c = xs[0].shape
    k = len(xs)
    dev = xs[0].device
    ys = [torch.empty_like(x) for x in xs]
    ab = torch.empty((k, 4, n, c), device=dev, dtype=torch.float32)
    assert ws is not None or fused_mask == 0
    hws = (C.c_int32 * k)(*[x.shape[1] * x.shape[2] for x in xs])
    onepass = fused_mask == 0 and GN_ONEPASS_FWD and gn_onepass_ok(xs[0], groups)
    if onepass:
        sync = _gn1p_sync(dev, k, n)
        ws1 = _gn1p_ws(dev, k, hws, n, c, groups, False)       # a NAMED tensor: a temporary would be freed (and handed out again) before the launch
        try:
            _lib.call("osd_groupnorm_relu_fwd_levels_onepass", k, _ptr_array(xs), _ptr_array(ys), hws, _p(gamma), _p(beta), _p(ab),
                      _p(ws1), _p(sync), n, c, groups, float(eps), _dt(xs[0]), _stream())
        except _lib.OsdError as e:
            if getattr(e, "code", 0) != -2:       # OSD_ERR_UNSUPPORTED: a map too large for one resident job — the two launches below
                raise
            onepass = False
    if not onepass:
        if ws is None:
            ws = torch.empty((k * n * GN_SPLITS * groups * 2,), device=dev, dtype=torch.float32)
        _lib.call("osd_groupnorm_relu_fwd_levels_fused", k, _ptr_array(xs), _ptr_array(ys), hws, _p(gamma), _p(beta), _p(ab),
                  _p(ws), n, c, groups, float(eps), _dt(xs[0]), int(fused_mask), _stream())
    for x, y in zip(xs, ys):
        _rec("gn_relu", x=x, gamma=gamma, beta=beta, groups=groups, eps=float(eps), out=y)
    return ys, ab


def gn_bwd_ws_numel(k, n, c, groups=32):
    """floats of osd_groupnorm_relu_bwd_levels' ws for k levels: the group sums, then the d gamma / d beta partials"""
    return k * n * GN_SPLITS * (groups * 2 + 2 * c)


def gn_bwd_ws_parts(ws, k, n, c, groups=32):
    """-> per level (group-sum block, d gamma / d beta block) views of ws: what osd_conv2d_fwd_multi_gn accumulates into"""
    a = n * GN_SPLITS * groups * 2
    b = n * GN_SPLITS * 2 * c
    return [(ws[l * a:(l + 1) * a], ws[k * a + l * b:k * a + (l + 1) * b]) for l in range(k)]


def groupnorm_relu_bwd_levels(us, dts, ab, gamma, beta, dgamma, dbeta, groups=32, ws=None, fused_mask=0, conv_db=None):
    """ws / fused_mask: the levels whose bit is set had their sums accumulated into (the zeroed) ws by the conv that wrote dts
    (conv2d_multi(gnb=...)); their statistics pass is skipped.
    conv_db [c] fp32 (optional; not with fused_mask): += the bias gradient of the conv that produced us (sum of du over levels,
    images, pixels), from sums the two passes gather anyway (osd_groupnorm_relu_bwd_levels_convbias).  Precedence: a non-None
    conv_db always selects the two-launch conv-bias kernels (the one-pass backward kernel does not emit that sum); TrainEngine
    therefore leaves conv_db off when OSD_GN_ONEPASS asks for the one-pass backward."""
    n, _, _, c = us[0].shape
    k = len(us)
    dev = us[0].device
    dus = [torch.empty_like(u) for u in us]
    assert ws is not None or fused_mask == 0
    hws = (C.c_int32 * k)(*[u.shape[1] * u.shape[2] for u in us])
    if conv_db is not None:
        assert fused_mask == 0 and ws is None
        ws = torch.empty((k * n * GN_SPLITS * (groups * 2 + 3 * c),), device=dev, dtype=torch.float32)
        _lib.call("osd_groupnorm_relu_bwd_levels_convbias", k, _ptr_array(us), _ptr_array(dts), _ptr_array(dus), hws, _p(ab),
                  _p(gamma), _p(beta), _p(ws), _p(dgamma), _p(dbeta), _p(conv_db), n, c, groups, _dt(us[0]), _stream())
    else:
        onepass = fused_mask == 0 and GN_ONEPASS_BWD and gn_onepass_ok(us[0], groups)
        if onepass:
            sync = _gn1p_sync(dev, k, n)
            ws1 = _gn1p_ws(dev, k, hws, n, c, groups, True)
            try:
                _lib.call("osd_groupnorm_relu_bwd_levels_onepass", k, _ptr_array(us), _ptr_array(dts), _ptr_array(dus), hws, _p(ab),
                          _p(gamma), _p(beta), _p(ws1), _p(sync), _p(dgamma), _p(dbeta), n, c, groups, _dt(us[0]), _stream())
            except _lib.OsdError as e:
                if getattr(e, "code", 0) != -2:
                    raise
                onepass = False
    if conv_db is None and not onepass:
        if ws is None:
            ws = torch.empty((gn_bwd_ws_numel(k, n, c, groups),), device=dev, dtype=torch.float32)
        _lib.call("osd_groupnorm_relu_bwd_levels_fused", k, _ptr_array(us), _ptr_array(dts), _ptr_array(dus), hws, _p(ab), _p(gamma),
                  _p(beta), _p(ws), _p(dgamma), _p(dbeta), n, c, groups, _dt(us[0]), int(fused_mask), _stream())
    _rec("gn_relu_bwd", us=list(us), dts=list(dts), gamma=gamma, beta=beta, groups=groups, dgamma=dgamma, dbeta=dbeta, outs=dus,
         conv_db=conv_db)
    return dus


# ---- second-stage ROI box head ----
def roi_pool_levels(feats, scales, boxes, counts, pool, sampling_ratio, out=None, want_levels=False):
    """Pooler.forward (modeling/poolers.py:93-124): feats = NHWC level maps, boxes [N,R,4] fp32 xyxy, counts [N] int32 or
    None -> [N*R, pool, pool, C] in the maps' dtype (zero rows past counts[image])."""
    _chk_dev(boxes, counts, *feats)
    n, r, _ = boxes.shape
    c = feats[0].shape[-1]
    k = len(feats)
    if out is None:
        out = torch.empty((n * r, pool, pool, c), device=boxes.device, dtype=feats[0].dtype)
    lv = torch.empty((n * r,), device=boxes.device, dtype=torch.int32) if want_levels else None
    xs = (C.c_void_p * k)(*[f.data_ptr() for f in feats])
    hs = (C.c_int32 * k)(*[f.shape[1] for f in feats])
    ws = (C.c_int32 * k)(*[f.shape[2] for f in feats])
    sc = (C.c_float * k)(*[float(s) for s in scales])
    assert all(f.shape[0] == n and f.shape[-1] == c and f.is_contiguous() for f in feats)
    _lib.call("osd_roi_pool_levels", k, xs, hs, ws, sc, _ptr(boxes.contiguous()), _p(counts), _p(out), n, c, r, pool,
              sampling_ratio, out.shape[-1], _p(lv), _dt(out), _stream())
    return (out, lv) if want_levels else out


def groupnorm_act_rois(x, gamma, beta, groups=32, eps=1e-5, slope=0.2, addend=None, rois_per_add=1, add_stride=1,
                       add_offset=0, out=None):
    """LeakyReLU(slope)(GroupNorm(groups, C)(x [+ addend map of the ROI's image / shot])) on [R,7,7,C] ROI maps."""
    _chk_dev(x, gamma, beta, addend)
    r, h, w, c = x.shape
    if out is None:
        out = torch.empty_like(x)
    if addend is not None:
        assert addend.dtype == x.dtype and addend.shape[1:] == x.shape[1:] and addend.is_contiguous()
    _lib.call("osd_groupnorm_act_rois", _p(x), _p(addend), _p(gamma), _p(beta), _p(out), r, h * w, c, groups,
              float(eps), float(slope), rois_per_add, add_stride, add_offset, _dt(x), _stream())
    return out


def box_decode(pred, rois, counts, reg_weights, img_h, img_w, score_thresh, want_raw=False, img_hw=None):
    """pred [S, N*R, P] (cols 0..1 logits, 2..9 deltas), rois [N,R,4] -> scores [N,R] (-1 = dropped), boxes [N,R,4]
    (+ the selected logits [N*R,2] and deltas [N*R,8] in fp32 when want_raw)."""
    _chk_dev(pred, rois, counts)
    s, m, p = pred.shape
    n, r, _ = rois.shape
    assert m == n * r and pred.is_contiguous()
    dev = pred.device
    scores = torch.empty((n, r), device=dev, dtype=torch.float32)
    boxes = torch.empty((n, r, 4), device=dev, dtype=torch.float32)
    lo = torch.empty((m, 2), device=dev, dtype=torch.float32) if want_raw else None
    ro = torch.empty((m, 8), device=dev, dtype=torch.float32) if want_raw else None
    rw = (C.c_float * 4)(*[float(v) for v in reg_weights])
    _lib.call("osd_box_decode", _p(pred), _ptr(rois.contiguous()), _p(counts), _p(scores), _p(boxes), _p(lo),
              _p(ro), n, r, s, p, rw, float(img_h), float(img_w), _p(img_hw), float(score_thresh), _dt(pred), _stream())
    return (scores, boxes, lo, ro) if want_raw else (scores, boxes)


def box_match_sample(boxes, counts, gt_boxes, gt_count, keys, batch_per_image, positive_fraction, iou_thresh, reg_weights,
                     gt_labels=None, want_all=False):
    """FastRCNNLossComputation.subsample on the device (osd_box_match_sample): boxes [N,P,4] (ground truth appended),
    counts [N], gt_boxes [N,G,4], gt_count [N], keys [N,P] fp32 uniform randoms -> sampled boxes [N,S,4], labels [N,S]
    int32, regression targets [N,S,4], proposal index [N,S] int32, counts [N] int32 (+ per-proposal labels / matches)."""
    _chk_dev(boxes, counts, gt_boxes, gt_count, keys)
    n, p, _ = boxes.shape
    s = int(batch_per_image)
    dev = boxes.device
    sb = torch.empty((n, s, 4), device=dev, dtype=torch.float32)
    sl = torch.empty((n, s), device=dev, dtype=torch.int32)
    st = torch.empty((n, s, 4), device=dev, dtype=torch.float32)
    si = torch.empty((n, s), device=dev, dtype=torch.int32)
    sc = torch.empty((n,), device=dev, dtype=torch.int32)
    al = torch.empty((n, p), device=dev, dtype=torch.int32) if want_all else None
    am = torch.empty((n, p), device=dev, dtype=torch.int32) if want_all else None
    rw = (C.c_float * 4)(*[float(v) for v in reg_weights])
    assert keys.shape == (n, p) and keys.dtype == torch.float32
    _lib.call("osd_box_match_sample", _ptr(boxes.contiguous()), _p(counts), _ptr(gt_boxes.contiguous().float()), _p(gt_count),
              _p(gt_labels), _ptr(keys.contiguous()), n, p, gt_boxes.shape[1], s, float(positive_fraction), float(iou_thresh),
              rw, _p(sb), _p(sl), _p(st), _p(si), _p(sc), _p(al), _p(am), _stream())
    return (sb, sl, st, si, sc, al, am) if want_all else (sb, sl, st, si, sc)


def box_loss(pred, labels, targets, s_count, n, rois_per_image, w_cls, w_box, grad_stride=0):
    """loss.py:306-381 ('ce_loss') x the weights of box_head.py:193-194.  pred [M, stride] (cols 0..1 logits, 2..9 deltas)
    -> losses [3] = (classification, box regression, valid rows) and, with grad_stride, d_pred [M, grad_stride]."""
    m = n * rois_per_image
    pred2 = pred.reshape(m, -1)
    losses = torch.empty((3,), device=pred.device, dtype=torch.float32)
    d = torch.empty((m, grad_stride), device=pred.device, dtype=pred.dtype) if grad_stride else None
    _lib.call("osd_box_loss", _p(pred2), _p(labels), _p(targets), _p(s_count), n, rois_per_image, pred2.shape[1],
              float(w_cls), float(w_box), _p(losses), _p(d), int(grad_stride), _dt(pred), _stream())
    return losses, d


def groupnorm_act_rois_bwd(x, gamma, beta, dy, dgamma, dbeta, groups=32, eps=1e-5, slope=0.2, addend=None, rois_per_add=1,
                           add_stride=1, add_offset=0):
    """Backward of groupnorm_act_rois: -> dx; dgamma / dbeta [C] fp32 accumulated."""
    r, h, w, c = x.shape
    dx = torch.empty_like(x)
    ws = torch.empty((r * 2 * c,), device=x.device, dtype=torch.float32)
    _lib.call("osd_groupnorm_act_rois_bwd", _p(x), _p(addend), _p(gamma), _p(beta), _ptr(dy.contiguous()), _p(dx), _p(ws),
              _p(dgamma), _p(dbeta), r, h * w, c, groups, float(eps), float(slope), rois_per_add, add_stride, add_offset,
              _dt(x), _stream())
    return dx


def rois_sum(x, n, rois_per_image):
    """[n * rois_per_image, ...] -> [n, ...]: the sum over each image's ROIs."""
    elems = x[0].numel()
    out = torch.empty((n,) + tuple(x.shape[1:]), device=x.device, dtype=x.dtype)
    _lib.call("osd_rois_sum", _p(x), _p(out), n, rois_per_image, elems, _dt(x), _stream())
    return out


def roi_pool_levels_bwd(shapes, scales, boxes, counts, dy, pool, sampling_ratio):
    """Backward of roi_pool_levels: dy [N*R, pool, pool, C] -> fp32 level maps [N, H_l, W_l, C] (zeroed here, then atomics)."""
    n, r, _ = boxes.shape
    c = dy.shape[-1]
    k = len(shapes)
    gxs = [torch.zeros((n, h, w, c), device=dy.device, dtype=torch.float32) for (h, w) in shapes]
    hs = (C.c_int32 * k)(*[h for h, _ in shapes])
    ws = (C.c_int32 * k)(*[w for _, w in shapes])
    sc = (C.c_float * k)(*[float(v) for v in scales])
    _lib.call("osd_roi_pool_levels_bwd", k, _ptr_array(gxs), hs, ws, sc, _ptr(boxes.contiguous()), _p(counts), _ptr(dy.contiguous()),
              n, c, r, pool, sampling_ratio, dy.shape[-1], _dt(dy), _stream())
    return gxs


def append_gt_boxes(boxes, scores, counts, gt_boxes, gt_count):
    """add_gt_proposals (fcos/inference.py:139-160): boxes [N,P,4], scores [N,P], counts [N] + gt_boxes [N,G,4], gt_count [N]
    -> boxes [N,P+G,4], scores [N,P+G], counts [N]."""
    _chk_dev(boxes, scores, counts, gt_boxes, gt_count)
    n, cap, _ = boxes.shape
    g = gt_boxes.shape[1]
    dev = boxes.device
    ob = torch.empty((n, cap + g, 4), device=dev, dtype=torch.float32)
    os_ = torch.empty((n, cap + g), device=dev, dtype=torch.float32)
    oc = torch.empty((n,), device=dev, dtype=torch.int32)
    _lib.call("osd_append_gt_boxes", _ptr(boxes.contiguous()), _ptr(scores.contiguous()), _p(counts),
              _ptr(gt_boxes.contiguous().float()), _p(gt_count), _p(ob), _p(os_), _p(oc), n, cap, g, _stream())
    return ob, os_, oc


def proposals_sort_nms(keys, boxes, max_count, levels, topn, thresh, max_keep, cuda_semantics=False, workspace=None,
                       head_hint=0, depth_out=None):
    """rank_sort_gather + nms_sorted in one call that ranks only the head of the score order (osd_proposals_sort_nms).
    keys [N,T] fp32 (dropped = -1), boxes [N,T,4] -> boxes [N,max_keep,4], scores [N,max_keep] (descending), counts [N]."""
    _chk_dev(keys, boxes)
    n, total = keys.shape
    dev = keys.device
    need = _lib.load().osd_proposals_workspace_bytes(n, total, max_count, max_keep)
    if workspace is None or workspace.numel() * workspace.element_size() < need:
        workspace = torch.empty((max(need, 8) // 8 + 1,), device=dev, dtype=torch.int64)
    ob = torch.zeros((n, max_keep, 4), device=dev, dtype=torch.float32)
    os_ = torch.zeros((n, max_keep), device=dev, dtype=torch.float32)
    oc = torch.empty((n,), device=dev, dtype=torch.int32)
    if levels:
        lo = (C.c_int32 * len(levels))(*[l for l, _ in levels])
        lc = (C.c_int32 * len(levels))(*[c for _, c in levels])
        nl = len(levels)
    else:
        lo, lc, nl = None, None, 0
    _lib.call("osd_proposals_sort_nms_hint", _p(keys), _p(boxes), n, total, max_count, lo, lc, nl, int(topn), float(thresh),
              int(cuda_semantics), max_keep, int(head_hint), _p(workspace), _p(ob), _p(os_), _p(oc), _p(depth_out),
              _stream())
    return ob, os_, oc
