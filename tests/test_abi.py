"""CPU-side checks of the drop-in boundary: libmgcr_hip.so loads, exports every symbol that
include/mgcr.h declares, and refuses to compute without a GPU (no fallback)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "mgcr.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(mgcr_[a-z0-9_]+)\s*\(", txt)))


def test_header_and_binding_agree():
    from mgpreconditionedgcr_amd import _lib
    assert header_symbols() == _lib.exported_symbols()


def test_library_exports_every_declared_symbol():
    from mgpreconditionedgcr_amd import _lib
    L = _lib.lib()  # resolves every symbol or raises
    for name in header_symbols():
        assert hasattr(L, name), name
    assert L.mgcr_version().startswith(b"mgcr-hip")


def test_no_silent_cpu_fallback():
    """Without a usable GPU every compute entry point must fail loudly."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from mgpreconditionedgcr_amd import _lib
    L = _lib.lib()
    assert L.mgcr_init(0) != 0
    assert b"no CPU fallback" in L.mgcr_last_error() or b"HIP" in L.mgcr_last_error()
    h = C.c_void_p()
    assert L.mgcr_vec_create(16, C.byref(h)) == 2  # MGCR_ERR_NO_DEVICE
    with pytest.raises(_lib.MgcrError):
        _lib.init(0)


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under the product package may import, include,
    link or load it (comments may mention it)."""
    pkg = os.path.join(ROOT, "mgpreconditionedgcr_amd")
    bad = re.compile(r"(import\s+oracle|from\s+oracle|#include\s*[\"<][^\">]*oracle|libmgcr_oracle|orc_[a-z_]+\s*\()")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")) or f == "Makefile":
                src = open(os.path.join(dirpath, f), errors="replace").read()
                assert not bad.search(src), f
