"""GPU: the ctypes stub INTEGRATION.md tells a reference maintainer to add (maskrcnn_benchmark/layers/_osd.py) is
executed VERBATIM from the document (only the library path is substituted) and checked against the oracle, so the
documented binding cannot rot."""
import os
import re

import numpy as np
import pytest
import torch

from oracle import hotpath_ref as orc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _stub_namespace():
    from oneshotdet_amd import _lib
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    block = re.search(r"```python\n(.*?)```", text, re.S).group(1)
    assert "/path/to/oneshotdet_amd/lib/liboneshotdet_hip.so" in block
    block = block.replace("/path/to/oneshotdet_amd/lib/liboneshotdet_hip.so", _lib.LIB_PATH)
    ns = {}
    exec(compile(block, "INTEGRATION.md", "exec"), ns)
    return ns


def test_documented_roi_align_stub_matches_the_oracle():
    ns = _stub_namespace()
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 16, 20, 24, generator=g)
    rois = torch.tensor([[0, 1.5, 2.0, 17.0, 15.5], [1, 0.0, 0.0, 23.0, 19.0], [1, 4.2, 3.3, 9.9, 12.1]])
    y = ns["roi_align_forward"](x.cuda(), rois.cuda(), 0.5, 3, 3, 2).cpu()
    ref = orc.roi_align(x, rois, 0.5, 3, 3, 2)
    torch.testing.assert_close(y, ref, rtol=1e-5, atol=1e-5)


def test_documented_focal_loss_stub_matches_the_oracle():
    ns = _stub_namespace()
    g = torch.Generator().manual_seed(1)
    logits = torch.randn(257, 1, generator=g) * 3
    targets = (torch.rand(257, generator=g) > 0.7).int()
    out = ns["sigmoid_focalloss_forward"](logits.cuda(), targets.cuda(), 1, 2.0, 0.25).cpu()
    ref = orc.sigmoid_focal_loss_cuda_formula(logits, targets, 2.0, 0.25)
    torch.testing.assert_close(out, ref, rtol=1e-5, atol=1e-6)


def test_documented_nms_stub_matches_the_oracle():
    ns = _stub_namespace()
    rng = np.random.RandomState(3)
    xy = rng.rand(900, 2).astype(np.float32) * 200
    boxes = np.concatenate([xy, xy + rng.rand(900, 2).astype(np.float32) * 90 + 2], 1)
    scores = rng.rand(900).astype(np.float32)
    keep = ns["nms"](torch.from_numpy(boxes).cuda(), torch.from_numpy(scores).cuda(), 0.6).cpu().numpy()
    assert np.array_equal(keep, orc.nms(boxes, scores, 0.6, cuda_semantics=True))


def test_documented_roi_align_backward_stub_matches_oracle_autograd():
    ns = _stub_namespace()
    g = torch.Generator().manual_seed(2)
    x = torch.randn(2, 8, 12, 10, generator=g, requires_grad=True)
    rois = torch.tensor([[0, 1.0, 2.0, 17.0, 15.5], [1, 0.0, 0.0, 19.0, 23.0]])
    ref = orc.roi_align(x, rois, 0.5, 2, 2, 2)
    w = torch.randn(ref.shape, generator=g)
    (ref * w).sum().backward()
    gx = ns["roi_align_backward"](w.cuda(), rois.cuda(), 0.5, 2, 2, 2, 8, 12, 10, 2).cpu()
    torch.testing.assert_close(gx, x.grad, rtol=1e-5, atol=1e-6)


def test_documented_conv_argument_order_is_the_headers():
    """INTEGRATION.md section 4 quotes osd_conv2d_fwd's parameter list: it must be the header's."""
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    hdr = open(os.path.join(ROOT, "include", "oneshotdet_hip.h")).read()
    m = re.search(r"int osd_conv2d_fwd\((.*?)\);", hdr, re.S)
    names = [a.strip().split()[-1].lstrip("*") for a in m.group(1).replace("\n", " ").split(",")]
    doc = re.search(r"`osd_conv2d_fwd\((.*?)\)`", text).group(1)
    assert [a.strip() for a in doc.split(",")] == ["desc" if n == "d" else n for n in names]


def test_training_example_runs_checkpoints_and_resumes(tmp_path):
    """examples/train.py: synthetic dataset samples -> reference-named transforms on device images -> collate into the stem
    input -> both-stage training steps -> reference-format checkpoints -> resume (momentum and iteration restored)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = str(tmp_path / "run")
    cmd = [sys.executable, os.path.join(root, "examples", "train.py"), "--batch", "2", "--out", out, "--checkpoint-period", "2"]
    r = subprocess.run(cmd + ["--iters", "3"], capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "loss_classifier" in r.stdout and os.path.exists(os.path.join(out, "model_0000003.pth"))
    r = subprocess.run(cmd + ["--iters", "5", "--resume"], capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "resumed at iteration 3" in r.stdout and os.path.exists(os.path.join(out, "model_0000005.pth"))
    import torch
    ck = torch.load(os.path.join(out, "model_0000005.pth"), map_location="cpu", weights_only=False)
    assert ck["iteration"] == 5 and "momentum_buffer" in ck["optimizer"] and "rpn.head.cls_tower.0.weight" in ck["model"]
