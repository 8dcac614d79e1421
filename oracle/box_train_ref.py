"""ORACLE — test infrastructure only.  CPU restatement of the TRAINING path of the reference's second-stage few-shot ROI box
head for the config of record ('concat', no negative support, 'ce_loss', 2 classes, class-specific
regression, BATCH_SIZE_PER_IMAGE 128, POSITIVE_FRACTION 0.25, FG = BG IoU threshold 0.5).

Paths relative to /root/reference/maskrcnn_benchmark/:
  modeling/roi_heads/box_head/loss.py:44-141     match_targets_to_proposals / prepare_targets
  modeling/roi_heads/box_head/loss.py:234-301    subsample
  modeling/roi_heads/box_head/loss.py:306-381    __call__ (gt_label == -1, 'ce_loss')
  modeling/matcher.py:52-83                      Matcher.__call__ (high = low = 0.5, allow_low_quality_matches False)
  modeling/balanced_positive_negative_sampler.py:19-62
  modeling/box_coder.py:21-50                    BoxCoder.encode, weights (10, 10, 5, 5)
  structures/boxlist_ops.py:221-256              boxlist_iou ("+1" areas)
  layers/smooth_l1_loss.py:5-15
  modeling/roi_heads/box_head/box_head.py:100-203  ROIBoxHead.forward in training: the losses are returned from INSIDE the
                                                 loop over query shots, so only the first query of every image is used;
                                                 weights 5 (classification) and 2.5 (box regression) :193-194

Randomness: the reference draws `torch.randperm(n)[:k]` among the positives / negatives.  The restatement takes uniform
random KEYS per proposal and keeps the k smallest of each class (ties: lower index) — the same thing with randperm(n) :=
argsort(keys of the class members, stable).  tests/golden/make_golden.py::gen_box_train_case runs the REAL reference's
subsample + ROIBoxHead in train mode with torch.randperm replaced by exactly that and asserts agreement (sampled rows, labels,
regression targets, both losses, parameter and feature gradients) before writing tests/golden/boxtrain_*.npz.
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import box_head_ref as obh

BATCH_PER_IMAGE = 128          # config/defaults.py:201
POSITIVE_FRACTION = 0.25       # :203
IOU_THRESH = 0.5               # :190,193
W_CLS, W_BOX = 5.0, 2.5        # box_head.py:193-194


def boxlist_iou(gt, props):
    """boxlist_ops.py:221-256 -> [G, P] float32."""
    a1 = (gt[:, 2] - gt[:, 0] + 1) * (gt[:, 3] - gt[:, 1] + 1)
    a2 = (props[:, 2] - props[:, 0] + 1) * (props[:, 3] - props[:, 1] + 1)
    lt = torch.max(gt[:, None, :2], props[:, :2])
    rb = torch.min(gt[:, None, 2:], props[:, 2:])
    wh = (rb - lt + 1).clamp(min=0)
    inter = wh[:, :, 0] * wh[:, :, 1]
    return inter / (a1[:, None] + a2 - inter)


def encode(reference_boxes, proposals, weights=obh.REG_WEIGHTS):
    """box_coder.py:21-50."""
    ew = proposals[:, 2] - proposals[:, 0] + 1
    eh = proposals[:, 3] - proposals[:, 1] + 1
    ecx = proposals[:, 0] + 0.5 * ew
    ecy = proposals[:, 1] + 0.5 * eh
    gw = reference_boxes[:, 2] - reference_boxes[:, 0] + 1
    gh = reference_boxes[:, 3] - reference_boxes[:, 1] + 1
    gcx = reference_boxes[:, 0] + 0.5 * gw
    gcy = reference_boxes[:, 1] + 0.5 * gh
    wx, wy, ww, wh = weights
    return torch.stack((wx * (gcx - ecx) / ew, wy * (gcy - ecy) / eh, ww * torch.log(gw / ew), wh * torch.log(gh / eh)), dim=1)


def match_labels(props, gt, gt_labels=None):
    """loss.py:44-141 for one image -> (labels [P] int64: 0 background / matched label, matched [P] int64: box or -1)."""
    q = boxlist_iou(gt, props)
    vals, matches = q.max(dim=0)
    matches = matches.clone()
    matches[vals < IOU_THRESH] = -1
    lab = (torch.ones(len(gt), dtype=torch.int64) if gt_labels is None else gt_labels.to(torch.int64))[matches.clamp(min=0)]
    lab = lab.clone()
    lab[matches == -1] = 0
    return lab, matches


def sample(labels, keys, batch=BATCH_PER_IMAGE, fraction=POSITIVE_FRACTION):
    """balanced_positive_negative_sampler.py:19-62 with randperm(n) := argsort(keys[members], stable) -> sampled proposal
    indices in ascending order (torch.nonzero(pos | neg), loss.py:292), and the two permutations the reference would draw."""
    pos = torch.nonzero(labels >= 1).squeeze(1)
    neg = torch.nonzero(labels == 0).squeeze(1)
    num_pos = min(pos.numel(), int(batch * fraction))
    num_neg = min(neg.numel(), batch - num_pos)
    perm1 = torch.from_numpy(np.argsort(keys[pos].numpy(), kind="stable"))
    perm2 = torch.from_numpy(np.argsort(keys[neg].numpy(), kind="stable"))
    chosen = torch.cat([pos[perm1[:num_pos]], neg[perm2[:num_neg]]])
    return torch.sort(chosen)[0], perm1, perm2


def subsample(props, gt, keys, gt_labels=None):
    """loss.py:234-301 for one image -> dict(index, boxes, labels, targets) of the sampled rows."""
    lab, matches = match_labels(props, gt, gt_labels)
    idx, _, _ = sample(lab, keys)
    tg = encode(gt[matches.clamp(min=0)], props)
    return dict(index=idx, boxes=props[idx], labels=lab[idx], targets=tg[idx], all_labels=lab, all_matched=matches)


def losses(class_logits, box_regression, labels, targets):
    """loss.py:306-381 (gt_label == -1, 'ce_loss', class-specific boxes) with the weights of box_head.py:193-194."""
    cls = F.cross_entropy(class_logits, labels)
    pos = torch.nonzero(labels > 0).squeeze(1)
    cols = 4 * labels[pos][:, None] + torch.tensor([0, 1, 2, 3])
    d = box_regression[pos[:, None], cols] - targets[pos]
    n = d.abs()
    box = torch.where(n < 1.0, 0.5 * n ** 2, n - 0.5).sum() / labels.numel()
    return W_CLS * cls, W_BOX * box


def box_train_forward(feats, query_feats, sampled, query_sizes, sd, shots=1):
    """ROIBoxHead.forward in training on ALREADY sampled proposals (list of dicts from subsample, equal counts per image:
    poolers.py:80).  query_feats holds B*S queries; only the first of every image reaches the loss (box_head.py:123-203).
    Returns (loss_classifier, loss_box_reg, class_logits, box_regression)."""
    B = len(sampled)
    qf = [q.view(B, shots, *q.shape[1:])[:, 0] for q in query_feats]
    qs = [query_sizes[i * shots] for i in range(B)]
    logits, reg, _ = obh.box_head_logits(feats, qf, [s["boxes"] for s in sampled], qs, sd)
    labels = torch.cat([s["labels"] for s in sampled])
    targets = torch.cat([s["targets"] for s in sampled])
    lc, lb = losses(logits, reg, labels, targets)
    return lc, lb, logits, reg
