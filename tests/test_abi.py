"""CPU: the C-ABI library builds, loads, exports every symbol include/auralis_amd.h declares, and fails loudly
without a GPU (no CPU fallback on the product path)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "auralis_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(aur_[a-z_0-9]+)\s*\(", src)))


def test_header_and_binding_agree():
    from auralis_amd import _lib
    assert _declared_symbols() == sorted(_lib.EXPORTS)


def test_library_exports_every_declared_symbol(lib_path):
    lib = ctypes.CDLL(lib_path)
    for name in _declared_symbols():
        assert hasattr(lib, name), name


def test_version_and_error_string(lib_path):
    from auralis_amd import _lib
    lib = _lib.load_library()
    assert lib.aur_version() == 1
    assert isinstance(lib.aur_last_error(), bytes)


def test_null_arguments_are_rejected(lib_path):
    from auralis_amd import _lib
    lib = _lib.load_library()
    assert lib.aur_engine_create(None, 0, None) == -1
    assert b"null argument" in lib.aur_last_error()
    assert lib.aur_step(None, None, None) == -1


def test_no_gpu_means_loud_failure(lib_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from auralis_amd._lib import AurError, NativeEngine
    with pytest.raises(AurError) as ei:
        NativeEngine(n_layer=1, max_seqs=1)
    assert "no CPU fallback" in str(ei.value) or "HIP" in str(ei.value)


def test_product_package_does_not_import_oracle():
    pkg = os.path.join(ROOT, "auralis_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", txt, flags=re.M), f


def test_gemm_tile_orders_are_bijections(lib_path):
    """aur_dbg_gemm_tile_map evaluates the kernel's own workgroup -> tile function on the host: the XCD-aware order, the
    plain order (weight-tile count not a multiple of 8) and the grouped experiment must each hit every tile exactly once."""
    import ctypes as C

    import numpy as np
    lib = ctypes.CDLL(lib_path)            # host-only entry point: no HIP device needed
    lib.aur_dbg_gemm_tile_map.argtypes = [C.c_int32] * 4 + [C.POINTER(C.c_int32)]
    cases = [(48, 4, 2, 0), (16, 16, 2, 0), (17, 4, 2, 0), (64, 1, 142, 0), (64, 1, 142, 8), (48, 1, 142, 8), (16, 1, 142, 8),
             (16, 1, 7, 8), (64, 1, 9, 4), (8, 1, 5, 2), (17, 1, 142, 8)]
    for gx, gy, gz, group in cases:
        n = gx * gy * gz
        out = np.zeros(3 * n, dtype=np.int32)
        assert lib.aur_dbg_gemm_tile_map(gx, gy, gz, group, out.ctypes.data_as(C.POINTER(C.c_int32))) == 0
        t = out.reshape(n, 3)
        assert t[:, 0].min() >= 0 and t[:, 0].max() < gx and t[:, 1].max() < gy and t[:, 2].max() < gz
        assert len({tuple(r) for r in t.tolist()}) == n, (gx, gy, gz, group)
        if (gx * gy) % 8 == 0:       # ids 8 apart share an XCD: every weight tile must stay on one XCD
            xcd_of = {}
            for L, (nt, sl, mt) in enumerate(t.tolist()):
                assert xcd_of.setdefault((nt, sl), L & 7) == (L & 7)
