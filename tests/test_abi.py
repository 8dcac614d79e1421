"""CPU: the C-ABI library loads and exports every symbol include/oneshotdet_hip.h declares; the ctypes binding lists
the same set (no compute calls without a GPU)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "oneshotdet_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(osd_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def lib_path():
    from oneshotdet_amd import build
    return build.build_library(verbose=False)


def test_header_and_binding_agree():
    from oneshotdet_amd import _lib
    assert declared_symbols() == sorted(_lib.SIGNATURES.keys())


def test_library_exports_every_declared_symbol(lib_path):
    import torch  # noqa: F401  resolves libamdhip64.so.7 to the runtime torch ships
    lib = ctypes.CDLL(lib_path)
    for name in declared_symbols():
        assert hasattr(lib, name), name
    lib.osd_abi_version.restype = ctypes.c_int
    from oneshotdet_amd import _lib
    assert lib.osd_abi_version() == _lib.ABI_VERSION == 4


def test_ops_fail_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from oneshotdet_amd import _lib, model, ops, spec, synth
    with pytest.raises(_lib.OsdError):
        model.HotPathEngine(synth.make_state_dict(spec.hot_path_shapes()))
    with pytest.raises(_lib.OsdError):
        ops.correlate(torch.zeros(1, 2, 2, 8), torch.zeros(1, 8))


def test_invalid_arguments_return_error_codes(lib_path):
    """Error behaviour of the boundary: negative return code + message, never a crash (no GPU needed: argument checks
    run before any launch)."""
    from oneshotdet_amd import _lib
    lib = _lib.load()
    d = _lib.ConvDesc()
    d.cout = 3
    rc = lib.osd_conv2d_fwd(ctypes.byref(d), None, None, None, None, None, None, None, None, None)
    assert rc == -1 and b"null" in lib.osd_last_error_string()
    rc = lib.osd_correlate_fwd(None, None, None, 1, 1, 8, 0, None)
    assert rc == -1
    assert lib.osd_correlate_fwd(None, None, None, 0, 1, 8, 0, None) == 0     # empty batch: no-op
    assert lib.osd_nms_workspace_bytes(2, 130) == 2 * 192 * 3 * 8 + (1 + 8) * 8      # mask rows padded to 64 + per-image flags
