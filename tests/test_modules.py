"""The nn.Module façades keep the reference's extension-point signatures and state_dict names:
CPU part = names / shapes / buffers vs the dump of the real model; GPU part = the contracts of the reference's own
tests/test_backbones.py:38-51 and parity with the engine / fixtures."""
import json
import os

import numpy as np
import pytest
import torch

import golden_utils as gu
from oneshotdet_amd import modules, spec, synth


def test_module_state_dicts_carry_the_reference_names():
    ref = json.load(open(os.path.join(gu.GOLDEN_DIR, "state_dict_keys.json")))["shapes"]
    bb = modules.build_backbone()
    head = modules.FCOSHead()
    got = {"backbone." + k: list(v.shape) for k, v in bb.state_dict().items()}
    got.update({"rpn.head." + k: list(v.shape) for k, v in head.state_dict().items()})
    want = {k: v for k, v in ref.items() if k.startswith("backbone.") or k.startswith("rpn.head.")}
    assert got == want
    # FrozenBN tensors are buffers, conv weights parameters (layers/batch_norm.py:12-17)
    buffers = {n for n, _ in bb.named_buffers()}
    assert "body.stem.bn1.running_var" in buffers and "body.layer1.0.downsample.1.weight" in buffers
    assert "body.stem.conv1.weight" in {n for n, _ in bb.named_parameters()}
    assert bb.out_channels == 256
    # a sub-state-dict of the reference loads unchanged
    sd = synth.make_state_dict(spec.hot_path_shapes())
    missing, unexpected = bb.load_state_dict({k[len("backbone."):]: torch.from_numpy(v) for k, v in sd.items()
                                              if k.startswith("backbone.")}, strict=True)
    assert not missing and not unexpected
    assert torch.equal(bb.state_dict()["fpn.fpn_inner4.bias"], torch.from_numpy(sd["backbone.fpn.fpn_inner4.bias"]))


def test_boxlist_mirrors_the_reference_class():
    bl = modules.BoxList(torch.zeros(3, 4), (160, 128))
    bl.add_field("scores", torch.ones(3))
    assert len(bl) == 3 and bl.size == (160, 128) and bl.mode == "xyxy" and bl.fields() == ["scores"]
    assert bl.has_field("scores") and not bl.has_field("labels")
    with pytest.raises(ValueError):
        modules.BoxList(torch.zeros(3, 5), (1, 1))


@pytest.mark.gpu
def test_backbone_and_head_modules_follow_the_reference_contracts():
    """tests/test_backbones.py:38-51: `out_channels` and one [N, out_channels, ., .] map per level; FCOSHead.forward
    returns three lists; values equal the engine's."""
    from oneshotdet_amd import model, ops
    sd = synth.make_state_dict(spec.hot_path_shapes())
    bb = modules.build_backbone().cuda()
    bb.load_state_dict({k[len("backbone."):]: torch.from_numpy(v) for k, v in sd.items() if k.startswith("backbone.")})
    N, H, W = 2, 224, 256
    x = torch.from_numpy(synth.make_images("mod.x", N, H, W, seed=2)).cuda()
    out = bb(x)
    assert len(out) == 5
    for lvl, (t, (h, w)) in enumerate(zip(out, spec.level_sizes(H, W))):
        assert t.shape == (N, bb.out_channels, h, w)
    eng = model.HotPathEngine(sd, dtype=torch.float32)
    ref = model.run_backbone(eng.backbone, x, torch.float32)
    for a, b in zip(out, ref):
        assert torch.equal(a.permute(0, 2, 3, 1), b)
    head = modules.FCOSHead().cuda()
    head.load_state_dict({k[len("rpn.head."):]: torch.from_numpy(v) for k, v in sd.items() if k.startswith("rpn.head.")})
    logits, reg, ctr = head(list(out))
    assert len(logits) == len(reg) == len(ctr) == 5
    eo = model.run_head(eng.head, ref)
    for l in range(5):
        assert logits[l].shape[1] == 1 and reg[l].shape[1] == 4 and ctr[l].shape[1] == 1
        assert torch.equal(logits[l][:, 0], eo[l][0][..., 0]) and torch.equal(reg[l].permute(0, 2, 3, 1), eo[l][1][..., :4])
    # reloading other weights repacks
    sd2 = synth.make_state_dict(spec.hot_path_shapes(), seed=3)
    bb.load_state_dict({k[len("backbone."):]: torch.from_numpy(v) for k, v in sd2.items() if k.startswith("backbone.")})
    assert not torch.equal(bb(x)[0], out[0])


@pytest.mark.gpu
def test_rpn_module_and_detector_return_boxlists():
    """FCOSModule.forward(images: ImageList, features, targets) -> (list[BoxList], {}) against the reference's proposals;
    OneShotDetector.forward(images, images_supp, target_ids=...) -> list[BoxList] with scores and labels against the
    second-stage fixture."""
    from oneshotdet_amd import layers
    name = "small"
    B, H, W, S, qh, qw = gu.CASES[name]
    img, q = gu.case_inputs(name)
    sd = synth.make_state_dict(spec.full_model_shapes())
    det = modules.OneShotDetector(sd, dtype=torch.float32)
    res = det(torch.from_numpy(img), torch.from_numpy(q), target_ids=[7] * B)
    f = gu.load("box_%s.npz" % name)
    assert len(res) == B and isinstance(res[0], modules.BoxList) and res[0].size == (W, H)
    assert bool((res[0].get_field("labels") == 7).all())
    got_b, got_s = res[0].bbox.cpu().numpy(), res[0].get_field("scores").cpu().numpy()
    assert gu.match_boxes(f["detections.0.boxes"], f["detections.0.scores"], got_b, got_s) >= 0.95
    # rpn façade on the engine's own correlated features
    eng = det.engine
    feats, qfeats, pooled, combined = eng.forward_features(torch.from_numpy(img).cuda(), torch.from_numpy(q).cuda())
    rpn = modules.FCOSModule().cuda().eval()
    rpn.head.load_state_dict({k[len("rpn.head."):]: torch.from_numpy(v) for k, v in sd.items() if k.startswith("rpn.head.")})
    boxes, losses = rpn(layers.to_image_list(torch.from_numpy(img).cuda()), [t.permute(0, 3, 1, 2) for t in combined])
    assert losses == {} and len(boxes) == B
    c = gu.load("case_%s.npz" % name)
    assert gu.match_boxes(c["proposals.0.boxes"], c["proposals.0.scores"], boxes[0].bbox.cpu().numpy(),
                          boxes[0].get_field("scores").cpu().numpy()) >= 0.99
    with pytest.raises(RuntimeError):
        rpn.train()(layers.to_image_list(torch.from_numpy(img).cuda()), [t.permute(0, 3, 1, 2) for t in combined])
