"""Drop-in counterparts of the reference's native-op wrappers (`maskrcnn_benchmark.layers`): same names, argument
meaning and error behaviour, NCHW logical layout at this level, backed by liboneshotdet_hip.so.

  nms(boxes, scores, thresh)                 layers/nms.py:5 -> _C.nms (csrc/nms.h:10-28)
  ROIAlign(output_size, spatial_scale, sampling_ratio)(input, rois)      layers/roi_align.py:50-68
  SigmoidFocalLoss(gamma, alpha)(logits, targets)                        layers/sigmoid_focal_loss.py:57-71
  ImageList / to_image_list(tensors, size_divisible)                     structures/image_list.py:8-73
"""
import math

import torch
from torch import nn

from . import ops


def _require_cuda(t, what):
    if not t.is_cuda:
        # reference: AT_ERROR("Not compiled with GPU support") / "Not implemented on the CPU" (csrc/nms.h:22, ROIAlign.h:44)
        raise RuntimeError("%s: oneshotdet_amd implements the GPU (MI355X) path only; got a %s tensor" % (what, t.device))


def nms(boxes, scores, nms_thresh, cuda_semantics=True, max_keep=None):
    """boxes [N,4] fp32 xyxy, scores [N] -> int64 indices of kept boxes, ascending in the original order (what
    csrc/cuda/nms.cu:127-130 returns).  Suppression when IoU > thresh (CUDA rule) or >= thresh (CPU rule).  One C-ABI
    call (osd_nms); max_keep (not in the reference): stop after that many survivors in score order."""
    _require_cuda(boxes, "nms")
    n = boxes.shape[0]
    if n == 0:
        return torch.empty((0,), dtype=torch.int64, device=boxes.device)
    if boxes.dim() != 2 or boxes.shape[1] != 4 or scores.shape[0] != n:
        raise RuntimeError("nms: boxes must be [N,4] and scores [N]")
    if max_keep is not None and max_keep < n:
        keys = scores.float().reshape(1, n).contiguous()
        bs, ss, idx, cnt = ops.rank_sort_gather(keys, boxes.float().reshape(1, n, 4).contiguous(), n)
        ob, os_, op, oc = ops.nms_sorted(bs, ss, cnt, nms_thresh, max_keep, cuda_semantics=cuda_semantics)
        k = int(oc[0].item())
        return idx[0].long()[op[0, :k].long()].sort()[0]
    keep, count = ops.nms(boxes.float().contiguous(), scores.float().contiguous(), nms_thresh, cuda_semantics)
    return keep[:int(count.item())]     # the reference API returns a dynamically sized tensor: one host read


class _ROIAlign(torch.autograd.Function):
    """layers/roi_align.py:11-44: forward = _C.roi_align_forward, backward = _C.roi_align_backward (gradient w.r.t. the
    input only; the ROIs get None there too).  Logical NCHW in and out; the kernels work on NHWC."""

    @staticmethod
    def forward(ctx, input, rois, output_size, spatial_scale, sampling_ratio):
        rois = rois.contiguous().float()
        ctx.save_for_backward(rois)
        ctx.output_size, ctx.spatial_scale, ctx.sampling_ratio = output_size, spatial_scale, sampling_ratio
        ctx.input_shape, ctx.input_dtype = input.shape, input.dtype
        x = input.permute(0, 2, 3, 1).contiguous()       # no copy when input is channels_last
        y = ops.roi_align(x, rois, spatial_scale, output_size[0], output_size[1], sampling_ratio)
        return y.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, grad_output):
        rois, = ctx.saved_tensors
        b, c, h, w = ctx.input_shape
        gy = grad_output.float().permute(0, 2, 3, 1).contiguous()      # [R, ph, pw, C]
        gx = ops.roi_align_bwd(gy, rois, (b, h, w, c), ctx.spatial_scale, ctx.output_size[0], ctx.output_size[1],
                               ctx.sampling_ratio)
        return gx.permute(0, 3, 1, 2).to(ctx.input_dtype), None, None, None, None


class ROIAlign(nn.Module):
    def __init__(self, output_size, spatial_scale, sampling_ratio):
        super(ROIAlign, self).__init__()
        self.output_size = (output_size, output_size) if isinstance(output_size, int) else tuple(output_size)
        self.spatial_scale = spatial_scale
        self.sampling_ratio = sampling_ratio

    def forward(self, input, rois):
        """input [B,C,H,W] (any memory format; fp32 or bf16), rois [R,5] -> [R,C,ph,pw] fp32, differentiable w.r.t. input."""
        _require_cuda(input, "ROIAlign")
        return _ROIAlign.apply(input, rois, self.output_size, self.spatial_scale, self.sampling_ratio)

    def __repr__(self):
        return "ROIAlign(output_size=%s, spatial_scale=%s, sampling_ratio=%s)" % (
            self.output_size, self.spatial_scale, self.sampling_ratio)


class _SigmoidFocalLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, targets, gamma, alpha):
        ctx.save_for_backward(logits, targets)
        ctx.gamma, ctx.alpha = gamma, alpha
        return ops.sigmoid_focal_loss_fwd(logits, targets, gamma, alpha)

    @staticmethod
    def backward(ctx, d_loss):
        logits, targets = ctx.saved_tensors
        return ops.sigmoid_focal_loss_bwd(logits, targets, d_loss.contiguous(), ctx.gamma, ctx.alpha), None, None, None


class SigmoidFocalLoss(nn.Module):
    def __init__(self, gamma, alpha):
        super(SigmoidFocalLoss, self).__init__()
        self.gamma, self.alpha = gamma, alpha

    def forward(self, logits, targets):
        _require_cuda(logits, "SigmoidFocalLoss")
        return _SigmoidFocalLoss.apply(logits.float(), targets.int(), self.gamma, self.alpha).sum()


class ImageList(object):
    """structures/image_list.py:8-27: a zero-padded batch [B,C,H,W] plus every image's true (height, width)."""

    def __init__(self, tensors, image_sizes):
        self.tensors = tensors
        self.image_sizes = [tuple(int(v) for v in s) for s in image_sizes]

    def to(self, *args, **kwargs):
        return ImageList(self.tensors.to(*args, **kwargs), self.image_sizes)


def to_image_list(tensors, size_divisible=0):
    """structures/image_list.py:30-73 (R0 of SURVEY.md 8a).  An ImageList passes through; a 3-D / 4-D tensor is taken as
    is (every image has the tensor's size); a list of CHW tensors is zero-padded at the bottom / right to the largest
    height and width, rounded up to a multiple of `size_divisible` (BatchCollator, data/collate_batch.py:15-20, passes
    DATALOADER.SIZE_DIVISIBILITY = 32), and the true sizes are kept for clip_to_image and the query ROI boxes.
    Data movement only: torch slice copies on whatever device the tensors live on."""
    if isinstance(tensors, torch.Tensor) and size_divisible > 0:
        tensors = [tensors] if tensors.dim() == 3 else list(tensors)
    if isinstance(tensors, ImageList):
        return tensors
    if isinstance(tensors, torch.Tensor):
        if tensors.dim() == 3:
            tensors = tensors[None]
        if tensors.dim() != 4:
            raise AssertionError("to_image_list: expected a 3-D or 4-D tensor")
        return ImageList(tensors, [t.shape[-2:] for t in tensors])
    if isinstance(tensors, (tuple, list)):
        max_size = [max(s) for s in zip(*[img.shape for img in tensors])]
        if size_divisible > 0:
            max_size[1] = int(math.ceil(max_size[1] / size_divisible) * size_divisible)
            max_size[2] = int(math.ceil(max_size[2] / size_divisible) * size_divisible)
        batched = tensors[0].new_zeros((len(tensors),) + tuple(max_size))
        for img, pad in zip(tensors, batched):
            pad[:img.shape[0], :img.shape[1], :img.shape[2]].copy_(img)
        return ImageList(batched, [im.shape[-2:] for im in tensors])
    raise TypeError("Unsupported type for to_image_list: {}".format(type(tensors)))
