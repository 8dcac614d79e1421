"""ORACLE — test infrastructure only.  CPU restatement of every KERNEL LAUNCH of the hot path, one function per launch kind.

Only `tests/` may import this module (never the product).  Where oracle/hotpath_ref.py restates the reference's functions
end to end in fp32, this file restates what ONE launch of the HIP library computes from the operands it was actually given,
in plain PyTorch-CPU fp32, so that the bf16 engines — whose end-to-end difference to an fp32 reference is dominated by their
own 8-bit roundings — can be checked launch by launch to within ONE bf16 unit in the last place:

    tests/test_gpu_launch_replay.py switches on `oneshotdet_amd.trace.TRACE`, runs a real forward / training step, and for
    every recorded launch calls the function below with the engine's OWN input tensors (teacher forcing) and compares the
    result, rounded once to the output dtype, with what the kernel wrote.

Each function cites the reference arithmetic it stands for (paths relative to /root/reference/maskrcnn_benchmark/).  The
conv / GroupNorm / pooling arithmetic itself is ATen's (as in the reference, SURVEY.md 8c "third-party arithmetic"); this
module only adds WHERE a launch adds bias, residual, mask and activation and where it rounds.  It is pinned in two ways
(tests/test_oracle_golden.py, CPU): chained end to end with rounding off it reproduces oracle/hotpath_ref.py (itself pinned
to the reference's fixtures), and its gradient functions reproduce torch autograd of the same expression.
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import hotpath_ref as orc

ACT_NONE, ACT_RELU, ACT_EXP_SCALE = 0, 1, 2
RES_NONE, RES_SAME, RES_UP2X, RES_DOWN2X = 0, 1, 2, 3


def nchw(t):
    """NHWC tensor of any float dtype -> NCHW fp32 (values unchanged: bf16 -> fp32 is exact)."""
    return t.float().permute(0, 3, 1, 2).contiguous()


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def store(t, dtype):
    """What a kernel's final store does to an fp32 value: round-to-nearest-even to `dtype` (identity for float32)."""
    return t.to(dtype).float()


def unpack_weight(w, cout, stem=False):
    """Kernel-layout weights [rows >= cout][R][S][cin_k] (K contiguous; FrozenBN already folded in by the packer,
    layers/batch_norm.py:19-24) -> OIHW fp32.  Stem: [rows][7][32] = 7 filter rows x (8 pixels x 4 channels), the 8th pixel
    and the 4th channel are zero weights (resnet.py:332-337 on the zero-padded NHWC4 image)."""
    w = w.float()
    if stem:
        rows = w.shape[0]
        return w.view(rows, 7, 8, 4)[:cout].permute(0, 3, 1, 2).contiguous()           # [cout, 4, 7, 8]
    return w[:cout].permute(0, 3, 1, 2).contiguous()                                    # [cout, cin_k, R, S]


def conv_launch(x, w, bias, cout, r, s, stem=False, stride=1, pad=0, act=ACT_NONE, res=None, res_mode=RES_NONE, relu_in=False,
                act_scale=1.0, act_scale_dev=None, mask=None, x2=None, x2_stride=1, w2=None, out_hw=None):
    """osd_conv2d_fwd / one segment of osd_conv2d_fwd_multi, forward or data-gradient use alike:
    fp32 accumulate of x (*) w, + bias, + residual (same size: resnet.py:311-313; nearest 2x of a half-size map:
    fpn.py:57-60), x (mask > 0) (ReLU backward of the producer), activation (ReLU resnet.py:299-313 / exp(scale * y)
    fcos.py:95-97).  x2: second pixel source of a 1x1 conv whose K is [x | x2] (conv3 + downsample as one GEMM).
    All tensors NHWC as the kernels hold them; returns NHWC fp32 BEFORE the final rounding."""
    xin = nchw(x)
    if relu_in:
        xin = F.relu(xin)                                       # fpn.py:98: P7 = conv(relu(P6))
    wt = unpack_weight(w, cout, stem)
    b = bias.float()[:cout]
    if stem:
        y = F.conv2d(xin, wt, b, stride=2)                      # the packed image already carries the padding of 3
        y = y[:, :, :out_hw[0], :out_hw[1]]
    elif x2 is not None:
        c1 = x.shape[-1]
        if w2 is None:
            wa, wb = wt[:, :c1], wt[:, c1:c1 + x2.shape[-1]]
        else:
            wa, wb = wt[:, :c1], unpack_weight(w2, cout)[:, :x2.shape[-1]]
        x2in = nchw(x2)[:, :, ::x2_stride, ::x2_stride]
        y = F.conv2d(xin, wa) + F.conv2d(x2in, wb) + b.view(1, -1, 1, 1)
    else:
        y = F.conv2d(xin, wt[:, :x.shape[-1]], b, stride=stride, padding=pad)
    if res_mode == RES_SAME:
        y = y + nchw(res)[:, :cout]
    elif res_mode == RES_UP2X:
        y = y + F.interpolate(nchw(res)[:, :cout], scale_factor=2, mode="nearest")
    elif res_mode == RES_DOWN2X:
        y = y + nchw(res)[:, :cout, ::2, ::2]
    if mask is not None:
        y = torch.where(nchw(mask)[:, :cout] > 0, y, torch.zeros_like(y))
    if act == ACT_RELU:
        y = F.relu(y)
    elif act == ACT_EXP_SCALE:
        sc = float(act_scale_dev.float().reshape(-1)[0]) if act_scale_dev is not None else float(act_scale)
        y = torch.exp(y * sc)
    return nhwc(y)


def pack_image_launch(x, out_shape, pad_t, pad_l):
    """osd_pack_image: NCHW fp32 -> zero padded NHWC4 (structures/image_list.py:56-63 pads bottom / right; the stem conv's
    padding of 3, resnet.py:333, is materialised at the top / left)."""
    n, c, h, w = x.shape
    out = torch.zeros(out_shape, dtype=torch.float32)
    out[:, pad_t:pad_t + h, pad_l:pad_l + w, :3] = x.permute(0, 2, 3, 1)
    return out


def maxpool_launch(x):
    """resnet.py:336: max_pool2d(kernel 3, stride 2, padding 1)."""
    return nhwc(F.max_pool2d(nchw(x), kernel_size=3, stride=2, padding=1))


def roi_align_launch(x, rois, scale, ph, pw, sampling_ratio):
    """csrc/cuda/ROIAlign_cuda.cu:65-122 on an NHWC map -> [R, ph, pw, C] fp32."""
    return orc.roi_align(nchw(x), rois.float(), scale, ph, pw, sampling_ratio).permute(0, 2, 3, 1).contiguous()


def roi_align_bwd_launch(gy, rois, x_shape, scale, ph, pw, sampling_ratio):
    """csrc/cuda/ROIAlign_cuda.cu:178-254: the gradient of roi_align w.r.t. its input map (autograd of the restatement
    above: the tap weights are constants)."""
    n, h, w, c = x_shape
    xin = torch.zeros((n, c, h, w), dtype=torch.float32, requires_grad=True)
    y = orc.roi_align(xin, rois.float(), scale, ph, pw, sampling_ratio)
    y.backward(gy.float().permute(0, 3, 1, 2))
    return nhwc(xin.grad)


def shot_mean_launch(x, batch):
    """generalized_rcnn.py:100-104 (batch_pooling): mean over the shots of each target image."""
    d, c = x.shape
    return x.float().view(batch, d // batch, c).mean(dim=1)


def shot_mean_bwd_launch(gy, shots):
    b, c = gy.shape
    return (gy.float() / shots)[:, None, :].expand(b, shots, c).reshape(b * shots, c).contiguous()


def query_pool_launch(xs, rois, scales, batch, sampling_ratio):
    """the 1 x 1 ROIAlign of every query's whole-image box + the mean over the shots, per level (generalized_rcnn.py:20-52, 100-104)"""
    return [shot_mean_launch(roi_align_launch(x, rois, sc, 1, 1, sampling_ratio).reshape(x.shape[0], -1), batch) for x, sc in zip(xs, scales)]


def query_pool_bwd_launch(dqs, rois, shapes, scales, shots, sampling_ratio):
    out = []
    for dq, sh, sc in zip(dqs, shapes, scales):
        dv = shot_mean_bwd_launch(dq, shots)
        out.append(roi_align_bwd_launch(dv.view(-1, 1, 1, dv.shape[-1]), rois, tuple(sh), sc, 1, 1, sampling_ratio))
    return out


def correlate_launch(x, q):
    """generalized_rcnn.py:307-311: features * pooled.expand(...)."""
    return x.float() * q.float()[:, None, None, :]


def correlate_bwd_query_launch(g, feat):
    """d pooled[n][c] = sum over pixels of g * feat (the other factor of the product above)."""
    return (g.double() * feat.double()).sum(dim=(1, 2)).float()


def add_mask_launch(a, b=None, mask=None):
    y = a.float() if b is None else a.float() + b.float()
    if mask is not None:
        y = torch.where(mask.float() > 0, y, torch.zeros_like(y))
    return y


def scatter2x_launch(x, out_hw, mask=None, addend=None):
    """Data gradient of a stride-2 conv's sampling: dst[2i][2j] = src[i][j], zero elsewhere (+ addend, x (mask > 0))."""
    n, ho, wo, c = x.shape
    h, w = out_hw
    y = torch.zeros((n, h, w, c), dtype=torch.float32)
    y[:, 0:2 * ho:2, 0:2 * wo:2] = x.float()
    if addend is not None:
        y = y + addend.float()
    if mask is not None:
        y = torch.where(mask.float() > 0, y, torch.zeros_like(y))
    return y


def upsample2x_bwd_launch(inner, prev=None):
    """Backward of F.interpolate(scale_factor=2, mode='nearest') (fpn.py:57): the 2x2 sum (+ the map's own gradient)."""
    t = inner.float()
    y = t[:, 0::2, 0::2] + t[:, 0::2, 1::2] + t[:, 1::2, 0::2] + t[:, 1::2, 1::2]
    return y if prev is None else y + prev.float()


def gn_relu_launch(x, gamma, beta, groups, eps):
    """fcos.py:37-38: GroupNorm(32, C) then ReLU, on an NHWC map."""
    return nhwc(F.relu(F.group_norm(nchw(x), groups, gamma.float(), beta.float(), eps)))


def gn_relu_bwd_launch(u, dt, gamma, beta, groups, eps):
    """Backward of the above through autograd -> (du NHWC, dgamma, dbeta)."""
    uu = nchw(u).requires_grad_(True)
    g = gamma.float().clone().requires_grad_(True)
    b = beta.float().clone().requires_grad_(True)
    y = F.relu(F.group_norm(uu, groups, g, b, eps))
    y.backward(nchw(dt))
    return nhwc(uu.grad), g.grad, b.grad


def wgrad_launch(x, dy, r, s, stride, pad, cout, scale=None, want_bias=False):
    """Weight gradient of conv(x; w * scale) w.r.t. w (scale = the folded FrozenBN row scale, batch_norm.py:20) from the
    STORED activations x and output gradients dy: -> (dw [cout][r][s][cin], db [cout] or None)."""
    xin = nchw(x)
    g = nchw(dy)[:, :cout]
    cin = xin.shape[1]
    dw = torch.nn.grad.conv2d_weight(xin, (cout, cin, r, s), g, stride=stride, padding=pad)
    if scale is not None:
        dw = dw * scale.float().view(-1, 1, 1, 1)
    db = g.double().sum(dim=(0, 2, 3)).float() if want_bias else None
    return dw.permute(0, 2, 3, 1).contiguous(), db


def pred_gather_launch(dys):
    """osd_pred_dy_gather: G[q][tap * 4 + co] = dy[q - (tap / 3 - 1, tap % 3 - 1)][co] (co < 4, 64 columns, zero outside the map), the levels
    one after the other — the prediction convs' (fcos.py:50-61) output gradients laid out so that both their weight gradient and
    their data gradient are 1x1 problems.  A pure copy: exact."""
    rows = []
    for dy in dys:
        n, h, w, _ = dy.shape
        d = dy[..., :4].float()
        g = torch.zeros((n, h, w, 64), dtype=torch.float32)
        for tap in range(9):
            oy, ox = tap // 3 - 1, tap % 3 - 1
            y0, y1, x0, x1 = max(0, oy), h + min(0, oy), max(0, ox), w + min(0, ox)       # q with q - (oy, ox) inside the map
            if y1 > y0 and x1 > x0:
                g[:, y0:y1, x0:x1, tap * 4:tap * 4 + 4] = d[:, y0 - oy:y1 - oy, x0 - ox:x1 - ox, :]
        rows.append(g.reshape(-1, 64))
    return torch.cat(rows, 0)


def pred_dgrad_pack_launch(w, cout, cin):
    """osd_pred_dgrad_pack: the fp32 master [cout][3][3][cin] as the [cin][1][1][64] weights of the 1x1 data-gradient conv over G:
    column tap * 4 + co of row ci = w[co][tap][ci]; the other columns zero."""
    w = w.float().reshape(cout, 9, cin)
    wd = torch.zeros((cin, 64), dtype=torch.float32)
    for tap in range(9):
        for co in range(cout):
            wd[:, tap * 4 + co] = w[co, tap]
    return wd.view(cin, 1, 1, 64)


def fcos_loss_grad_launch(head_out, gt_boxes, gt_count, scales, gamma, alpha):
    """Phase 1 of osd_fcos_loss_levels: d loss / d (logit, centerness) and d loss / d (bbox_pred conv output) from the STORED
    head outputs, by autograd of the loss restated in hotpath_ref.fcos_loss (fcos/loss.py:213-276).  reg = exp(scale * x), so
    d/dx = d/dreg * reg * scale and d/dscale = sum d/dreg * reg * log(reg) / scale (fcos.py:95-97).
    head_out: [(cls_ctr [N,H,W,4], reg [N,H,W,4])] per level.  -> per level (d_cls_ctr [N,H,W,2], d_x [N,H,W,4]), d_scale_raw"""
    logits, ctrs, regs = [], [], []
    for cc, rg in head_out:
        c = nchw(cc)
        logits.append(c[:, 0:1].clone().requires_grad_(True))
        ctrs.append(c[:, 1:2].clone().requires_grad_(True))
        regs.append(nchw(rg)[:, :4].clone().requires_grad_(True))
    gts = [gt_boxes[i, :int(gt_count[i])].float().numpy() for i in range(gt_boxes.shape[0])]
    c, r, t, info = orc.fcos_loss(logits, regs, ctrs, gts, gamma=gamma, alpha=alpha, focal="cuda")
    (c + r + t).backward()
    outs, raws = [], []
    for lvl in range(len(head_out)):
        g_reg = regs[lvl].grad if regs[lvl].grad is not None else torch.zeros_like(regs[lvl])
        ds = g_reg * regs[lvl].detach()
        d_x = ds * float(scales[lvl])
        raws.append(float((ds.double() * regs[lvl].detach().double().log()).sum()))
        gl = logits[lvl].grad if logits[lvl].grad is not None else torch.zeros_like(logits[lvl])
        gc = ctrs[lvl].grad if ctrs[lvl].grad is not None else torch.zeros_like(ctrs[lvl])
        outs.append((nhwc(torch.cat([gl, gc], 1)), nhwc(d_x)))
    return outs, raws, (float(c), float(r), float(t), info["num_pos"])


# ---------------------------------------------------------------------------------------------------------------- comparison
def bf16_ulp(ref):
    """Spacing of bfloat16 at |ref| (8 significant bits: 2^(floor(log2 |ref|) - 7)); the smallest normal's spacing below it."""
    a = ref.abs().clamp_min(torch.finfo(torch.float32).tiny)
    return torch.exp2(torch.floor(torch.log2(a)) - 7.0)


def compare(got, ref, dtype, noise=1e-5):
    """got: what the kernel stored (any dtype), ref: the fp32 restatement before rounding.  Returns a dict:
    bf16 output: `worst_ulp` = max |got - round(ref)| in units of (one bf16 ulp at |ref| + noise * absmax(ref)) — a kernel
    that accumulates in fp32 in another order may land on the neighbouring bf16 value, never further — and `flips`, the
    fraction of elements that differ from round(ref) at all;  fp32 output: `worst_rel` = max |got - ref| / (1e-4 |ref| +
    noise * absmax)."""
    g = got.float()
    absmax = float(ref.abs().max()) if ref.numel() else 0.0
    if not torch.isfinite(g).all():
        return dict(ok=False, why="non-finite output", worst_ulp=float("inf"), worst_rel=float("inf"), flips=1.0, absmax=absmax)
    if dtype == torch.bfloat16:
        rr = store(ref, dtype)
        tol = bf16_ulp(ref) + noise * absmax
        d = (g - rr).abs()
        worst = float((d / tol).max()) if d.numel() else 0.0
        flips = float((d > 0).float().mean()) if d.numel() else 0.0
        return dict(ok=worst <= 1.0, worst_ulp=worst, flips=flips, absmax=absmax)
    tol = 1e-4 * ref.abs() + noise * absmax + 1e-30
    worst = float(((g - ref).abs() / tol).max()) if g.numel() else 0.0
    return dict(ok=worst <= 1.0, worst_rel=worst, flips=0.0, absmax=absmax)
