import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(autouse=True, scope="module")
def fresh_tuner_caches():
    """The per-shape kernel choices of `ops.tuning()` live in process-wide caches: a test module must not inherit what an
    earlier module's tuning left there (bit-for-bit comparisons between two batch sizes hold among kernels that sum K in one
    order; a cached choice for ONE of the two shapes breaks them, depending on the order the modules ran in)."""
    try:
        from oneshotdet_amd import tuner
    except Exception:      # noqa: BLE001  (library not built: the CPU tests that need it fail by themselves)
        yield
        return
    for cache in (tuner.ALGO_CACHE, tuner.SPLIT_CACHE, tuner.WGRAD_ALGO_CACHE):
        cache.clear()
    yield
