"""The C-ABI library must load and export every symbol include/ppcx.h declares (no compute calls here:
those need a GPU and live in the `-m gpu` tests)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "ppcx.h")).read()
    return sorted(set(re.findall(r"PPCX_API\s+[\w\s\*]+?\b(ppcx_\w+)\s*\(", txt)))


def test_header_declares_the_boundary():
    names = _declared()
    for must in ["ppcx_model_create", "ppcx_log_prob_grad", "ppcx_fit_nuts", "ppcx_fit_ppc", "ppcx_do_inference_C"]:
        assert must in names


def test_library_exports_every_declared_symbol():
    from ppcseq_amd import build
    lib_path = build.build()
    lib = ctypes.CDLL(lib_path)
    for name in _declared():
        assert hasattr(lib, name), f"{name} declared in include/ppcx.h but not exported by libppcx.so"
    lib.ppcx_version.restype = ctypes.c_int
    assert lib.ppcx_version() >= 100


def test_binding_lists_the_same_symbols():
    from ppcseq_amd import _lib
    assert sorted(_lib.EXPORTS) == _declared()


def test_argument_validation_needs_no_gpu():
    from ppcseq_amd import _lib
    lib = _lib.load()
    h = ctypes.c_void_p()
    rc = lib.ppcx_model_create(0, 0, 5, 2, 0, None, None, None, 5.6, 0, None, ctypes.byref(h))
    assert rc == -1 and b"G>=1" in lib.ppcx_last_error()


def test_product_does_not_import_oracle():
    """The shipped package must never route through oracle/ (only tests, smoke and the bench baseline may)."""
    pkg = os.path.join(ROOT, "ppcseq_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, re.M), f
                assert "ppc_oracle" not in txt and "libppc_oracle" not in txt, f
