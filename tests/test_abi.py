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
    assert lib.ppcx_version() == 400          # include/ppcx.h PPCX_VERSION


def test_test_hooks_are_not_in_the_shipped_library():
    """Fault injection, forced cell paths, the stand-in nccl provider and the kernel-level timing exist only in the testing
    build (tests/libppcx_testing.so, -DPPCX_TESTING); the product library exports none of it and carries none of the old
    environment switches."""
    from ppcseq_amd import build
    lib = ctypes.CDLL(build.build())
    for name in ("ppcx_testing_set", "ppcx_testing_set_nccl_provider", "ppcx_testing_bench_kernel", "ppcx_bench_gene_kernel"):
        assert not hasattr(lib, name), name
    blob = open(build.build(), "rb").read()
    for s in (b"PPCX_TEST_FAIL", b"PPCX_RCCL_LIB", b"PPCX_TWO_GROUP", b"PPCX_NO_TAIL_TIERS", b"PPCX_PLAN_IGNORE_TIERS", b"injected failure"):
        assert s not in blob, s
    tlib = ctypes.CDLL(build.build_testing())
    for name in ("ppcx_testing_set", "ppcx_testing_set_nccl_provider", "ppcx_testing_bench_kernel"):
        assert hasattr(tlib, name), name
    for name in _declared():                     # the testing build is the product plus the hooks
        assert hasattr(tlib, name), name


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


def test_guard_decision_of_the_gene_shard_ranks():
    """What the ranks of a gene-sharded run conclude at a poll from the max-reduced vector [rounds, -rounds, done, -done,
    -(error)] (ppcx_capi.hip comm_guard): agreement, a peer's error, their own error, disagreement on rounds or chains."""
    import numpy as np
    from ppcseq_amd import build
    lib = ctypes.CDLL(build.build())
    lib.ppcx_guard_decision.argtypes = [ctypes.POINTER(ctypes.c_double), ctypes.c_int]
    lib.ppcx_guard_decision.restype = ctypes.c_int

    def decide(vecs, local):
        red = np.max(np.array(vecs, float), axis=0)                        # what ncclMax leaves on every rank
        return lib.ppcx_guard_decision(red.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), local)

    def vec(rounds, done, err=0):
        return [rounds, -rounds, done, -done, -err if err else 0.0]
    assert decide([vec(64, 1), vec(64, 1)], 0) == 0                        # all agree
    assert decide([vec(64, 1), vec(96, 1)], 0) == -5                       # rounds issued differ: PPCX_ERR_STALL
    assert decide([vec(64, 1), vec(64, 2)], 0) == -5                       # chains done differ
    assert decide([vec(64, 0), vec(64, 0, err=-2)], 0) == -2               # a peer failed: its class
    assert decide([vec(64, 0, err=-3), vec(64, 0)], -3) == -3              # this rank failed: its own status
    assert decide([vec(64, 0, err=-2), vec(32, 0, err=-6)], -6) == -6      # both failed: each keeps its own


def test_r_shim_is_a_source_file_written_for_this_abi_version():
    """r/ppcx_do_inference.R (the .C() shim that replaces R/utilities.R:1482-1531) and r/zzz.R are source files of the
    repository; the version the shim passes as dims[1] is the header's PPCX_VERSION, and its .C() call names every argument
    of ppcx_do_inference_C in the header's order. (R is not installed here: the call itself is exercised by the plain-C host
    tests/c_host/dot_c_host.c on the GPU box.)"""
    hdr = open(os.path.join(ROOT, "include", "ppcx.h")).read()
    version = int(re.search(r"#define PPCX_VERSION (\d+)", hdr).group(1))
    shim = open(os.path.join(ROOT, "r", "ppcx_do_inference.R")).read()
    assert re.search(r"as\.integer\(c\(%dL," % version, shim), "the shim's dims[1] must be PPCX_VERSION"
    call = shim[shim.index('.C("ppcx_do_inference_C"'):]
    names = re.findall(r"^\s+(\w+)\s*=", call[:call.index("if (out$status")], re.M)
    assert names == ["dims", "counts", "X", "expo", "excl", "reals", "ci", "slope", "counts_rng", "status", "errbuf", "errlen"]
    proto = hdr[hdr.index("PPCX_API void ppcx_do_inference_C"):]
    proto = proto[:proto.index(";")]
    assert proto.count(",") + 1 == len(names)
    assert "dyn.load" in open(os.path.join(ROOT, "r", "zzz.R")).read()
    host = open(os.path.join(ROOT, "tests", "c_host", "dot_c_host.c")).read()
    assert "int dims[33] = {%d," % version in host
