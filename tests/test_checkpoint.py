"""CPU: weight interchange with the reference's `.pth` checkpoints (oneshotdet_amd/checkpoint.py; reference
utils/checkpoint.py:33-103, utils/model_serialization.py:10-80)."""
import os

import numpy as np
import pytest
import torch

from oneshotdet_amd import checkpoint, spec, synth


def _sd(shapes):
    return {k: torch.from_numpy(v) for k, v in synth.make_state_dict(shapes).items()}


def test_round_trip_in_reference_format(tmp_path):
    sd = _sd(spec.full_model_shapes())
    p = checkpoint.save_checkpoint(str(tmp_path / "model_0000010.pth"), sd, iteration=10)
    assert open(os.path.join(str(tmp_path), "last_checkpoint")).read() == p
    raw = torch.load(p, map_location="cpu", weights_only=False)
    assert set(raw.keys()) == {"model", "iteration"} and list(raw["model"].keys()) == list(sd.keys())
    got, extras = checkpoint.load_checkpoint(p)
    assert extras == {"iteration": 10} and list(got.keys()) == list(sd.keys())
    for k in sd:
        assert torch.equal(got[k], sd[k]), k


def test_ddp_prefix_and_suffix_matching(tmp_path):
    """A DistributedDataParallel checkpoint (`module.` on every key) loads; so does a file whose keys are only suffixes
    of the model's (model_serialization.py docstring), the longest suffix winning."""
    sd = _sd(spec.hot_path_shapes())
    p = str(tmp_path / "ddp.pth")
    torch.save({"model": {"module." + k: v for k, v in sd.items()}}, p)
    got, _ = checkpoint.load_checkpoint(p)
    assert all(torch.equal(got[k], sd[k]) for k in sd)
    exp = {"backbone.body.layer1.0.conv1.weight": (2, 2), "backbone.body.stem.conv1.weight": (3, 3)}
    loaded = {"conv1.weight": torch.ones(3, 3), "layer1.0.conv1.weight": torch.zeros(2, 2)}
    out, missing = checkpoint.align_state_dict(exp, loaded)
    assert not missing and out["backbone.body.layer1.0.conv1.weight"].shape == (2, 2)
    assert out["backbone.body.stem.conv1.weight"].shape == (3, 3)


def test_first_stage_only_file_and_errors(tmp_path):
    sd = _sd(spec.hot_path_shapes())
    p = str(tmp_path / "rpn_only.pth")
    torch.save(sd, p)                                      # bare state_dict, no "model" wrapper (checkpoint.py:164)
    got, _ = checkpoint.load_checkpoint(p)
    assert list(got.keys()) == list(spec.hot_path_shapes().keys())
    with pytest.raises(KeyError):
        checkpoint.load_checkpoint(p, second_stage=True)
    full = _sd(spec.full_model_shapes())
    got, _ = checkpoint.load_checkpoint(p, second_stage=True, defaults=full)      # UNLOAD_KEYWORD-style fill
    assert np.array_equal(got["roi_heads.box.fc7.bias"].numpy(), full["roi_heads.box.fc7.bias"].numpy())
    bad = dict(sd)
    bad["rpn.head.cls_logits.bias"] = torch.zeros(3)
    torch.save({"model": bad}, p)
    with pytest.raises(ValueError):
        checkpoint.load_checkpoint(p)


def _fake_c2_r50(rng):
    """Blob names of a Detectron R-50.pkl (conv1 + AffineChannel, res2..res5 bottlenecks, fc1000, momentum blobs)."""
    blobs = {"conv1_w": rng.randn(64, 3, 7, 7).astype(np.float32), "res_conv1_bn_s": rng.rand(64).astype(np.float32),
             "res_conv1_bn_b": rng.randn(64).astype(np.float32), "fc1000_w": rng.randn(1000, 2048).astype(np.float32),
             "fc1000_b": rng.randn(1000).astype(np.float32), "conv1_w_momentum": np.zeros((64, 3, 7, 7), np.float32)}
    cin = 64
    for si, nblocks in enumerate(spec.STAGE_BLOCKS):
        mid, cout = 64 * 2 ** si, 256 * 2 ** si
        for b in range(nblocks):
            p = "res%d_%d_" % (si + 2, b)
            if b == 0:
                blobs[p + "branch1_w"] = rng.randn(cout, cin, 1, 1).astype(np.float32)
                blobs[p + "branch1_bn_s"], blobs[p + "branch1_bn_b"] = rng.rand(cout).astype(np.float32), rng.randn(cout).astype(np.float32)
            for br, (ci, co, k) in (("branch2a", (cin, mid, 1)), ("branch2b", (mid, mid, 3)), ("branch2c", (mid, cout, 1))):
                blobs[p + br + "_w"] = rng.randn(co, ci, k, k).astype(np.float32)
                blobs[p + br + "_bn_s"], blobs[p + br + "_bn_b"] = rng.rand(co).astype(np.float32), rng.randn(co).astype(np.float32)
            cin = cout
    return blobs


def test_caffe2_resnet_pickle_fills_both_backbones(tmp_path):
    """utils/c2_model_loading.py:12-175 + the suffix alignment: `res4_5_branch2b_w` lands in layer3.5.conv2 of BOTH
    backbones, AffineChannel scale / bias become FrozenBN weight / bias, running statistics / FPN / head keep the
    defaults, momentum blobs and fc1000 are ignored."""
    import pickle
    blobs = _fake_c2_r50(np.random.RandomState(0))
    p = str(tmp_path / "R-50.pkl")
    with open(p, "wb") as f:
        pickle.dump({"blobs": blobs}, f, protocol=2)
    defaults = {k: torch.from_numpy(v) for k, v in synth.make_state_dict(spec.hot_path_shapes()).items()}
    sd = checkpoint.load_c2_resnet(p, defaults)
    assert list(sd.keys()) == list(spec.hot_path_shapes().keys())
    for bb in ("backbone.", "supp_backbone."):
        assert np.array_equal(sd[bb + "body.layer3.5.conv2.weight"].numpy(), blobs["res4_5_branch2b_w"])
        assert np.array_equal(sd[bb + "body.stem.conv1.weight"].numpy(), blobs["conv1_w"])
        assert np.array_equal(sd[bb + "body.stem.bn1.weight"].numpy(), blobs["res_conv1_bn_s"])
        assert np.array_equal(sd[bb + "body.layer2.0.downsample.1.bias"].numpy(), blobs["res3_0_branch1_bn_b"])
        assert np.array_equal(sd[bb + "body.layer4.2.bn3.weight"].numpy(), blobs["res5_2_branch2c_bn_s"])
        assert torch.equal(sd[bb + "body.layer1.0.bn1.running_var"], defaults[bb + "body.layer1.0.bn1.running_var"])
        assert torch.equal(sd[bb + "fpn.fpn_inner2.weight"], defaults[bb + "fpn.fpn_inner2.weight"])
    assert torch.equal(sd["rpn.head.cls_logits.bias"], defaults["rpn.head.cls_logits.bias"])
    assert checkpoint.translate_c2_resnet_name("res2_0_branch2a_w_momentum") is None
    assert checkpoint.translate_c2_resnet_name("res5_1_branch2c_bn_b") == "layer4.1.bn3.bias"
    del blobs["res3_1_branch2b_w"]
    with open(p, "wb") as f:
        pickle.dump(blobs, f, protocol=2)                      # bare dict, one blob missing
    with pytest.raises(KeyError):
        checkpoint.load_c2_resnet(p, defaults)
