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


def test_struct_fields_of_header_and_binding_agree():
    """Every struct of include/auralis_amd.h is mirrored field for field, in order, by the ctypes binding (a field added on one side
    only shifts every later member)."""
    from auralis_amd import _lib
    src = open(os.path.join(ROOT, "include", "auralis_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    for name in ("aur_config", "aur_tensor_desc", "aur_seq_desc", "aur_result", "aur_stats", "aur_cond_params"):
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (name, name), src, re.S).group(1)
        fields = [re.sub(r"\[\d+\]", "", d.strip().split()[-1].lstrip("*")) for d in body.split(";") if d.strip()]
        bound = [f for f, _ in getattr(_lib, name)._fields_]
        assert fields == bound, (name, fields, bound)


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
