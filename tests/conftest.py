import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle.oracle import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def bundled():
    """The reference's bundled example data (data/counts.rda) as exported by tests/golden/make_counts_fixture.py."""
    z = np.load(os.path.join(ROOT, "tests", "golden", "counts_bundled.npz"), allow_pickle=False)
    return {k: z[k] for k in z.files}


def bundled_test_config(bundled, checked=("SLC16A12", "CYP1A1", "ART3"), how_many_negative_controls=50):
    """The selection of tests/testthat/test-ppcSeq.R:11-24: three checked genes + the 50 least
    significant others (select_to_check_and_house_keeping, R/utilities.R:628-649), checked genes first."""
    genes = [str(g) for g in bundled["genes"]]
    idx_checked = [genes.index(g) for g in checked]
    others = [i for i in range(len(genes)) if i not in idx_checked]
    order = sorted(others, key=lambda i: bundled["PValue"][i])       # stable ascending p-value
    controls = set(order[-how_many_negative_controls:])
    sel = idx_checked + [i for i in others if i in controls]          # first-appearance order
    counts = bundled["value"][sel].astype(np.int32)
    label = bundled["Label"]
    X = np.stack([np.ones(len(label)), (label == sorted(set(label))[1]).astype(float)], axis=1)
    return counts, X, [genes[i] for i in sel], len(idx_checked)


@pytest.fixture(scope="session")
def emul():
    """CPU emulation harness built from the product's __host__ __device__ headers (tests/emul)."""
    import ctypes as C
    import subprocess
    here = os.path.join(ROOT, "tests", "emul")
    lib = os.path.join(here, "libppcx_emul.so")
    src = os.path.join(here, "ppcx_emul.cpp")
    hdrs = [os.path.join(ROOT, "ppcseq_amd", "csrc", h) for h in ("ppcx_math.h", "ppcx_disp.h", "ppcx_model.h", "ppcx_nuts.h", "ppcx_gene.h")]
    if not os.path.exists(lib) or os.path.getmtime(lib) < max(os.path.getmtime(p) for p in [src] + hdrs):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden", "-o", lib, src])
    return C.CDLL(lib)
