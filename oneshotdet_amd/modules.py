"""`nn.Module` façades with the reference's own extension-point signatures ("Python-level plugin API"):

  build_backbone(dtype)            -> module with `.out_channels`, forward(NCHW) -> tuple of per-level NCHW maps
                                      (modeling/backbone/backbone.py:51-72,98-103; contract of tests/test_backbones.py:38-51)
  FCOSHead.forward(list[Tensor])   -> (logits, bbox_reg, centerness) lists           (modeling/rpn/fcos/fcos.py:83-99)
  FCOSModule.forward(images: ImageList, features, targets=None) -> (list[BoxList], {})  (fcos.py:145-176, rpn/rpn.py:201)
  OneShotDetector.forward(images, images_supp, targets=None, device=None, target_ids=None) -> list[BoxList]
                                                                                       (detector/generalized_rcnn.py:226-332)

Parameters and buffers carry the REFERENCE's names and OIHW shapes (`body.stem.conv1.weight`, `fpn.fpn_inner2.bias`,
`cls_tower.0.weight`, `scales.0.scale`, FrozenBN `weight / bias / running_mean / running_var`, ...), so
`module.load_state_dict(reference_sub_state_dict)` works unchanged; the kernels' NHWC / K-contiguous, BN-folded packing
happens lazily at the first forward after a load.  Inference modules: they run the HIP path under no_grad and return
logically-NCHW tensors (channels-last memory).  Training goes through `train.TrainEngine`.
"""
import torch
from torch import nn

from . import layers, model, ops, spec


class BoxList(object):
    """structures/bounding_box.py:9-60: boxes [N,4] + image size (width, height) + named per-box fields."""

    def __init__(self, bbox, image_size, mode="xyxy"):
        bbox = torch.as_tensor(bbox, dtype=torch.float32)
        if bbox.ndimension() != 2 or bbox.size(-1) != 4:
            raise ValueError("bbox should be [N, 4], got {}".format(tuple(bbox.shape)))
        if mode not in ("xyxy", "xywh"):
            raise ValueError("mode should be 'xyxy' or 'xywh'")
        self.bbox, self.size, self.mode, self.extra_fields = bbox, image_size, mode, {}

    def add_field(self, field, field_data):
        self.extra_fields[field] = field_data

    def get_field(self, field):
        return self.extra_fields[field]

    def has_field(self, field):
        return field in self.extra_fields

    def fields(self):
        return list(self.extra_fields.keys())

    def __len__(self):
        return self.bbox.shape[0]

    def __repr__(self):
        return "BoxList(num_boxes={}, image_width={}, image_height={}, mode={})".format(len(self), self.size[0], self.size[1], self.mode)


class _PackedModule(nn.Module):
    """Registers tensors under reference names; packs them for the kernels on first use after a (re)load."""

    def __init__(self, shapes, prefix, dtype):
        super(_PackedModule, self).__init__()
        self._prefix, self.dtype, self._packed = prefix, dtype, None
        self._names = []
        for key, shape in shapes.items():
            assert key.startswith(prefix)
            name = key[len(prefix):]
            t = torch.zeros(tuple(shape), dtype=torch.float32)
            leaf = name.rsplit(".", 1)[-1]
            # FrozenBatchNorm2d keeps all four tensors as buffers (layers/batch_norm.py:12-17); everything else is a Parameter
            if ".bn" in name or "downsample.1." in name or leaf.startswith("running_"):
                self._register(name, t, buffer=True)
            else:
                self._register(name, nn.Parameter(t, requires_grad=False), buffer=False)
            self._names.append(name)

    def _register(self, dotted, value, buffer):
        mod = self
        parts = dotted.split(".")
        for p in parts[:-1]:
            if not hasattr(mod, p):
                mod.add_module(p, nn.Module())
            mod = getattr(mod, p)
        (mod.register_buffer if buffer else mod.register_parameter)(parts[-1], value)

    def _load_from_state_dict(self, *a, **k):
        self._packed = None
        return super(_PackedModule, self)._load_from_state_dict(*a, **k)

    def load_state_dict(self, *a, **k):
        self._packed = None
        return super(_PackedModule, self).load_state_dict(*a, **k)

    def _sd(self):
        sd = self.state_dict()
        return {self._prefix + k: v.detach() for k, v in sd.items()}


class ResNetFPNBackbone(_PackedModule):
    """Sequential(OrderedDict(body=ResNet, fpn=FPN)) of backbone.py:51-72 for R-50-FPN-RETINANET (P3..P7)."""

    def __init__(self, dtype=torch.float32):
        super(ResNetFPNBackbone, self).__init__(spec.backbone_shapes("backbone."), "backbone.", dtype)
        self.out_channels = spec.FPN_OUT

    def packed(self):
        if self._packed is None:
            self._packed = model.BackboneWeights(self._sd(), "backbone.", self.dtype)
        return self._packed

    def forward_nhwc(self, x):
        layers._require_cuda(x, "backbone")
        with torch.no_grad():
            return model.run_backbone(self.packed(), x.float().contiguous(), self.dtype)

    def forward(self, x):
        return tuple(t.permute(0, 3, 1, 2) for t in self.forward_nhwc(x))


def build_backbone(dtype=torch.float32):
    """registry.BACKBONES["R-50-FPN-RETINANET"](cfg) (backbone.py:51-72,98-103)."""
    return ResNetFPNBackbone(dtype)


class FCOSHead(_PackedModule):
    """modeling/rpn/fcos/fcos.py:12-99."""

    def __init__(self, dtype=torch.float32):
        super(FCOSHead, self).__init__(spec.fcos_head_shapes("rpn.head."), "rpn.head.", dtype)

    def packed(self):
        if self._packed is None:
            self._packed = model.HeadWeights(self._sd(), self.dtype)
        return self._packed

    def forward_nhwc(self, feats_nhwc):
        with torch.no_grad():
            return model.run_head(self.packed(), feats_nhwc)

    def forward(self, x):
        """x: list of NCHW-logical level maps -> (logits, bbox_reg, centerness) lists of NCHW-logical tensors."""
        feats = [t.permute(0, 2, 3, 1).contiguous().to(self.dtype) for t in x]     # no copy for channels-last inputs
        out = self.forward_nhwc(feats)
        logits = [c[..., 0:1].permute(0, 3, 1, 2) for c, _ in out]
        ctr = [c[..., 1:2].permute(0, 3, 1, 2) for c, _ in out]
        reg = [r[..., 0:4].permute(0, 3, 1, 2) for _, r in out]
        return logits, reg, ctr


def _boxlists(boxes, scores, counts, image_sizes, labels=None):
    """Device [N,K,4] / [N,K] / [N] -> list[BoxList] (one host read of the counts: the reference's API returns
    dynamically sized results)."""
    out = []
    cnt = counts.cpu().tolist()
    for i, k in enumerate(cnt):
        h, w = image_sizes[i]
        bl = BoxList(boxes[i, :k], (int(w), int(h)), mode="xyxy")
        bl.add_field("scores", scores[i, :k])
        if labels is not None:
            bl.add_field("labels", torch.full((k,), int(labels[i]), dtype=torch.int64, device=boxes.device))
        out.append(bl)
    return out


class FCOSModule(nn.Module):
    """build_rpn(cfg, in_channels) for FCOS_ON (rpn/rpn.py:201-206; fcos.py:102-207), eval path."""

    def __init__(self, dtype=torch.float32):
        super(FCOSModule, self).__init__()
        self.head = FCOSHead(dtype)

    def forward(self, images, features, targets=None):
        if self.training:
            raise RuntimeError("FCOSModule façade is the inference path; training runs through oneshotdet_amd.train.TrainEngine")
        images = layers.to_image_list(images)
        feats = [t.permute(0, 2, 3, 1).contiguous().to(self.head.dtype) for t in features]
        head_out = self.head.forward_nhwc(feats)
        h, w = images.tensors.shape[-2:]
        b, s, c = model.run_proposals(head_out, h, w, spec.PRE_NMS_TOP_N_TEST, spec.POST_NMS_TOP_N_TEST, spec.NMS_THRESH,
                                      image_sizes=images.image_sizes)
        return _boxlists(b, s, c, images.image_sizes), {}


class OneShotDetector(nn.Module):
    """GeneralizedRCNN (detector/generalized_rcnn.py:55-332) in eval mode: `backbone`, `supp_backbone`, `rpn` (and the
    `roi_heads.box.*` entries when present) under the reference's names, forward -> list[BoxList] with `scores` and
    `labels` (= target_ids[i]) like the reference returns."""

    def __init__(self, state_dict, dtype=torch.float32, device="cuda"):
        super(OneShotDetector, self).__init__()
        self.engine = model.HotPathEngine(state_dict, dtype=dtype, device=device)
        self.second_stage = self.engine.box_head is not None

    def state_dict(self, *a, **k):
        return dict(self.engine.sd)

    def forward(self, images, images_supp, targets=None, device=None, target_ids=None):
        images = layers.to_image_list(images)
        images_supp = layers.to_image_list(images_supp)
        dev = self.engine.device
        imgs = layers.ImageList(images.tensors.to(dev, torch.float32), images.image_sizes)
        qs = layers.ImageList(images_supp.tensors.to(dev, torch.float32), images_supp.image_sizes)
        with torch.no_grad():
            out = self.engine.detect(imgs, qs, second_stage=self.second_stage)
        if self.second_stage:
            d = out["detections"]
            ids = target_ids if target_ids is not None else [1] * imgs.tensors.shape[0]
            return _boxlists(d["boxes"], d["scores"], d["counts"], imgs.image_sizes, labels=ids)
        b, s, c = out["proposals"]
        return _boxlists(b, s, c, imgs.image_sizes)
