"""The drop-in boundary bound WITHOUT Python: tests/c_host/dot_c_host.c (gcc, dlopen, no HIP or C++ headers) calls
`ppcx_do_inference_C` with exactly the argument shapes of R's .C() -- what r/ppcx_do_inference.R does in place of
R/utilities.R:1482-1531 -- and runs the reference's testthat case (tests/testthat/test-ppcSeq.R:7-32) through both passes of
identify_outliers. The host runs as a fresh child process (it has not touched the GPU before) and must print the reference's
only known answer: tot_deleterious_outliers = 0 1 0, the outlier being CYP1A1 in sample 11165PP (count 5835)."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HERE = os.path.join(ROOT, "tests", "c_host")


def _host():
    exe, src = os.path.join(HERE, "dot_c_host"), os.path.join(HERE, "dot_c_host.c")
    if not os.path.exists(exe) or os.path.getmtime(exe) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O2", "-Wall", "-o", exe, src, "-ldl", "-lm"])
    return exe


@pytest.mark.parametrize("mode", ["vb", "nuts", "nuts-2-devices"])
def test_plain_c_host_reproduces_the_reference_known_answer(mode):
    """mode nuts-2-devices: dims names two devices (device 0 twice on this box) -- the chains of both passes are dealt to two host
    threads inside the library, as rstan::sampling deals them to `cores` workers (R/utilities.R:1500-1501)."""
    from ppcseq_amd import build
    lib = build.build()
    args = [_host(), lib, os.path.join(HERE, "bundled_53x21.txt"), "1"] + (["nuts"] if mode != "vb" else []) + (["2"] if mode == "nuts-2-devices" else [])
    p = subprocess.run(args, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr
    lines = p.stdout.strip().splitlines()
    assert lines[0] == "0 1 0", p.stdout                                   # tests/testthat/test-ppcSeq.R:26-30
    assert lines[1].startswith("outlier gene 2 sample 17 count 5835"), p.stdout     # CYP1A1, 11165PP


def test_c_host_reports_a_stale_abi_version_instead_of_reading_on():
    """A caller written for another argument layout (dims[0] != PPCX_VERSION) gets status PPCX_ERR_ARG and a message."""
    import ctypes as C
    from ppcseq_amd import build
    lib = C.CDLL(build.build())
    dims = (C.c_int * 33)(300, *([0] * 32))
    reals = (C.c_double * 6)()
    ci = (C.c_double * 4)()
    status, errlen = (C.c_int * 1)(7), (C.c_int * 1)(128)
    buf = C.create_string_buffer(128)
    errbuf = (C.c_char_p * 1)(C.addressof(buf))
    lib.ppcx_do_inference_C(dims, None, None, None, None, reals, ci, None, None, status, errbuf, errlen)
    assert status[0] == -1 and b"ABI version" in buf.value
