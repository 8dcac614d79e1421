"""Weight interchange with the reference's native checkpoints (`.pth` part).

Mirrors utils/checkpoint.py:33-103 (`{"model": state_dict, "optimizer": ..., "iteration": ...}` written by torch.save,
`last_checkpoint` tag file) and utils/model_serialization.py:10-80 (strip a DataParallel `module.` prefix, then give every
expected key the loaded key that is its LONGEST suffix), so a reference `model_XXXXXXX.pth` feeds
`HotPathEngine` / `TrainEngine` unchanged and `TrainEngine.state_dict()` goes back out in the reference's format.
The Caffe2 / Detectron `.pkl` route (utils/c2_model_loading.py:12-175; `MODEL.WEIGHT: catalog://ImageNetPretrained/MSRA/R-50`)
is `load_c2_resnet`: blob names are translated to the torchvision-style names the reference maps them to, and the same
suffix alignment then fills BOTH backbones (every `*.body.layer1.0.conv1.weight` ends with `layer1.0.conv1.weight`).
"""
import os
import pickle
import re
from collections import OrderedDict

import torch

from . import spec


def strip_prefix_if_present(state_dict, prefix="module."):
    """model_serialization.py:58-66: only when EVERY key carries the prefix."""
    if not state_dict or not all(k.startswith(prefix) for k in state_dict):
        return state_dict
    return OrderedDict((k.replace(prefix, ""), v) for k, v in state_dict.items())


def align_state_dict(expected_shapes, loaded):
    """model_serialization.py:10-55: for each expected key pick the loaded key that is a suffix of it, longest first.
    Returns (aligned {expected key: tensor}, missing expected keys).  Shapes are checked (the reference would fail later,
    inside nn.Module.load_state_dict)."""
    loaded = strip_prefix_if_present(loaded)
    loaded_keys = sorted(loaded.keys())
    out, missing = OrderedDict(), []
    for key in expected_shapes:
        best = None
        for lk in loaded_keys:
            if key.endswith(lk) and (best is None or len(lk) > len(best)):
                best = lk
        if best is None:
            missing.append(key)
            continue
        t = torch.as_tensor(loaded[best])
        if tuple(t.shape) != tuple(expected_shapes[key]):
            raise ValueError("%s: checkpoint entry %s has shape %s, expected %s"
                             % (key, best, tuple(t.shape), tuple(expected_shapes[key])))
        out[key] = t.detach().to(torch.float32).cpu()
    return out, missing


def load_checkpoint(path, second_stage=None, defaults=None):
    """Read a reference `.pth` (or a bare state_dict file) -> (state_dict under the reference's key names, extras).
    second_stage: True = require roi_heads.box.*, False = first stage only, None = take it when present.
    defaults: values for keys the file lacks (utils/checkpoint.py:107-115 keeps the model's own initialisation for
    FEW_SHOT.UNLOAD_KEYWORD modules); without it a missing key is an error."""
    data = torch.load(path, map_location="cpu", weights_only=False)
    if not isinstance(data, dict):
        raise ValueError("%s: not a checkpoint dictionary" % path)
    if "model" not in data:                                            # checkpoint.py:164-165
        data = {"model": data}
    loaded = data.pop("model")
    shapes = spec.hot_path_shapes()
    box = spec.box_head_shapes()
    probe = strip_prefix_if_present(loaded)
    has_box = any(k.endswith("box.fc6.weight") for k in probe)
    if second_stage or (second_stage is None and has_box):
        shapes.update(box)
    sd, missing = align_state_dict(shapes, loaded)
    for k in list(missing):
        if defaults is not None and k in defaults:
            sd[k] = torch.as_tensor(defaults[k]).to(torch.float32).cpu()
            missing.remove(k)
    if missing:
        raise KeyError("%s lacks %d expected entries, e.g. %s" % (path, len(missing), missing[:3]))
    return OrderedDict((k, sd[k]) for k in shapes), data


def save_checkpoint(path, state_dict, tag_last=True, **extras):
    """utils/checkpoint.py:33-50: {"model": state_dict, **extras} + the `last_checkpoint` tag file next to it."""
    data = {"model": OrderedDict((k, torch.as_tensor(v).detach().cpu()) for k, v in state_dict.items())}
    data.update(extras)
    d = os.path.dirname(os.path.abspath(path))
    os.makedirs(d, exist_ok=True)
    torch.save(data, path)
    if tag_last:
        with open(os.path.join(d, "last_checkpoint"), "w") as f:       # checkpoint.py:95-98
            f.write(path)
    return path


def save_training_checkpoint(path, engine, iteration, tag_last=True):
    """utils/checkpoint.py:33-50 as the trainer calls it (engine/trainer.py:111-119): model + optimizer + iteration.
    `optimizer` holds TrainEngine.optimizer_state_dict() (momentum buffers under reference names, steps taken, lr)."""
    return save_checkpoint(path, engine.state_dict(), tag_last=tag_last, optimizer=engine.optimizer_state_dict(),
                           iteration=int(iteration))


def resume_training(path, make_engine):
    """Load a checkpoint written by save_training_checkpoint: make_engine(state_dict) -> TrainEngine; its momentum and
    step count are restored.  Returns (engine, iteration)."""
    sd, extras = load_checkpoint(path)
    eng = make_engine(sd)
    if "optimizer" in extras and isinstance(extras["optimizer"], dict) and "momentum_buffer" in extras["optimizer"]:
        eng.load_optimizer_state_dict(extras["optimizer"])
    return eng, int(extras.get("iteration", 0))


_C2_BRANCH = {"branch2a": ("conv1", "bn1"), "branch2b": ("conv2", "bn2"), "branch2c": ("conv3", "bn3"),
              "branch1": ("downsample.0", "downsample.1")}
_C2_BLOB = re.compile(r"^res(\d)_(\d+)_(branch2a|branch2b|branch2c|branch1)(_bn)?_(w|s|b)$")


def translate_c2_resnet_name(name):
    """One Caffe2 ResNet blob name -> the name the reference gives it (utils/c2_model_loading.py:12-60: `res2_0_branch2a_w`
    -> `layer1.0.conv1.weight`, `res_conv1_bn_s` -> `bn1.weight`, `res3_0_branch1_bn_b` -> `layer2.0.downsample.1.bias`,
    ...; AffineChannel scale / bias become the FrozenBN weight / bias).  None for blobs the loader skips (momentum)."""
    if name.endswith("_momentum"):
        return None
    if name == "conv1_w":
        return "conv1.weight"
    if name == "conv1_b":
        return "conv1.bias"
    if name in ("res_conv1_bn_s", "conv1_bn_s"):
        return "bn1.weight"
    if name in ("res_conv1_bn_b", "conv1_bn_b"):
        return "bn1.bias"
    if name in ("fc1000_w", "pred_w"):
        return "fc1000.weight"
    if name in ("fc1000_b", "pred_b"):
        return "fc1000.bias"
    m = _C2_BLOB.match(name)
    if m is None:
        return name.replace("_", ".")          # anything else keeps the reference's basic `_` -> `.` renaming
    stage, block, branch, bn, kind = int(m.group(1)), int(m.group(2)), m.group(3), m.group(4), m.group(5)
    conv, norm = _C2_BRANCH[branch]
    if bn:
        leaf = {"s": "weight", "b": "bias"}.get(kind)
        if leaf is None:
            return None
        return "layer%d.%d.%s.%s" % (stage - 1, block, norm, leaf)
    leaf = {"w": "weight", "b": "bias"}.get(kind)
    return None if leaf is None else "layer%d.%d.%s.%s" % (stage - 1, block, conv, leaf)


def load_c2_resnet(path, defaults, second_stage=None):
    """A Detectron ResNet `.pkl` (dict of numpy blobs, optionally under "blobs"; pickled by Python 2: latin1) -> a full
    state_dict: the ResNet bodies of BOTH backbones come from the file, every other entry (FrozenBN running statistics —
    AffineChannel has none —, FPN, FCOS head, second stage) from `defaults`, as `DetectronCheckpointer.load` leaves the
    model's own initialisation for what the file lacks."""
    with open(path, "rb") as f:
        data = pickle.load(f, encoding="latin1")
    blobs = data["blobs"] if isinstance(data, dict) and "blobs" in data else data
    loaded = OrderedDict()
    for k in sorted(blobs.keys()):
        name = translate_c2_resnet_name(k)
        if name is not None:
            loaded[name] = torch.as_tensor(blobs[k])
    shapes = spec.hot_path_shapes()
    if second_stage or (second_stage is None and all(k in defaults for k in spec.box_head_shapes())):
        shapes.update(spec.box_head_shapes())
    body = OrderedDict((k, v) for k, v in shapes.items() if ".body." in k and not k.endswith(("running_mean", "running_var")))
    sd, missing = align_state_dict(body, loaded)
    if missing:
        raise KeyError("%s lacks %d ResNet entries, e.g. %s" % (path, len(missing), missing[:3]))
    out = OrderedDict()
    for k in shapes:
        out[k] = sd[k] if k in sd else torch.as_tensor(defaults[k]).to(torch.float32).cpu()
    return out
