"""Case lists and deterministic source images of the transforms fixtures, shared by tests/golden/make_golden.py (writer,
build container only) and the tests that read tests/golden/transforms.npz (they must regenerate the same uint8 inputs)."""
import numpy as np

from oneshotdet_amd import synth

TRANSFORM_CASES = [
    # name, (h, w) of the uint8 source, which pipeline ("target": 800 / 1200, "support": 200 / 400, "tiny": 48 / 80), flip
    ("landscape", (375, 500), "target", False),
    ("portrait_maxsize", (500, 333), "target", False),
    ("landscape_flip", (333, 500), "target", True),
    ("no_resize", (800, 1000), "target", False),
    ("downscale", (1200, 1600), "target", True),
    ("support_up", (127, 127), "support", False),
    ("support_wide_flip", (90, 160), "support", True),
    ("tiny_full", (37, 53), "tiny", False),
    ("tiny_full_flip_down", (120, 90), "tiny", True),
]
TRANSFORM_SIZES = {"target": (800, 1200), "support": (200, 400), "tiny": (48, 80)}


def transform_source(name, hw):
    """Deterministic uint8 RGB test image [h, w, 3]: smooth structure + hashed noise (so resampling differences show)."""
    h, w = hw
    u = synth.uniform01("transform." + name, h * w * 3, seed=5).reshape(h, w, 3)
    yy, xx = np.mgrid[0:h, 0:w]
    base = 0.5 + 0.5 * np.sin(yy[..., None] / 17.0 + np.array([0.0, 1.0, 2.0])) * np.cos(xx[..., None] / 23.0)
    return np.clip((0.6 * base + 0.4 * u) * 255.0, 0, 255).astype(np.uint8)
