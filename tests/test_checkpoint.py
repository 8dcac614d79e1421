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
