"""Weight interchange with the reference's native checkpoints (`.pth` part).

Mirrors utils/checkpoint.py:33-103 (`{"model": state_dict, "optimizer": ..., "iteration": ...}` written by torch.save,
`last_checkpoint` tag file) and utils/model_serialization.py:10-80 (strip a DataParallel `module.` prefix, then give every
expected key the loaded key that is its LONGEST suffix), so a reference `model_XXXXXXX.pth` feeds
`HotPathEngine` / `TrainEngine` unchanged and `TrainEngine.state_dict()` goes back out in the reference's format.
Not built: the Caffe2 `.pkl` route (utils/c2_model_loading.py: Detectron key renaming) — it needs the Detectron model
zoo, which is not reachable here.
"""
import os
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
