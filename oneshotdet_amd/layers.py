"""Drop-in counterparts of the reference's native-op wrappers (`maskrcnn_benchmark.layers`): same names, argument
meaning and error behaviour, NCHW logical layout at this level, backed by liboneshotdet_hip.so.

  nms(boxes, scores, thresh)                 layers/nms.py:5 -> _C.nms (csrc/nms.h:10-28)
  ROIAlign(output_size, spatial_scale, sampling_ratio)(input, rois)      layers/roi_align.py:50-68
  SigmoidFocalLoss(gamma, alpha)(logits, targets)                        layers/sigmoid_focal_loss.py:57-71
"""
import torch
from torch import nn

from . import ops


def _require_cuda(t, what):
    if not t.is_cuda:
        # reference: AT_ERROR("Not compiled with GPU support") / "Not implemented on the CPU" (csrc/nms.h:22, ROIAlign.h:44)
        raise RuntimeError("%s: oneshotdet_amd implements the GPU (MI355X) path only; got a %s tensor" % (what, t.device))


def nms(boxes, scores, nms_thresh, cuda_semantics=True, max_keep=None):
    """boxes [N,4] fp32 xyxy, scores [N] -> int64 indices of kept boxes, ascending in the original order (what
    csrc/cuda/nms.cu:127-130 returns).  Suppression when IoU > thresh (CUDA rule) or >= thresh (CPU rule)."""
    _require_cuda(boxes, "nms")
    n = boxes.shape[0]
    if n == 0:
        return torch.empty((0,), dtype=torch.int64, device=boxes.device)
    if boxes.dim() != 2 or boxes.shape[1] != 4 or scores.shape[0] != n:
        raise RuntimeError("nms: boxes must be [N,4] and scores [N]")
    keys = scores.float().reshape(1, n).contiguous()
    bs, ss, idx, cnt = ops.rank_sort_gather(keys, boxes.float().reshape(1, n, 4).contiguous(), n)
    max_keep = n if max_keep is None else max_keep
    ob, os_, op, oc = ops.nms_sorted(bs, ss, cnt, nms_thresh, max_keep, cuda_semantics=cuda_semantics)
    k = int(oc[0].item())           # the reference API returns a dynamically sized tensor: one host read
    kept_sorted_pos = op[0, :k].long()
    return idx[0].long()[kept_sorted_pos].sort()[0]


class ROIAlign(nn.Module):
    def __init__(self, output_size, spatial_scale, sampling_ratio):
        super(ROIAlign, self).__init__()
        self.output_size = (output_size, output_size) if isinstance(output_size, int) else tuple(output_size)
        self.spatial_scale = spatial_scale
        self.sampling_ratio = sampling_ratio

    def forward(self, input, rois):
        """input [B,C,H,W] (any memory format), rois [R,5] -> [R,C,ph,pw] fp32."""
        _require_cuda(input, "ROIAlign")
        x = input.permute(0, 2, 3, 1).contiguous()       # no copy when input is channels_last
        y = ops.roi_align(x, rois, self.spatial_scale, self.output_size[0], self.output_size[1], self.sampling_ratio)
        return y.permute(0, 3, 1, 2)

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
