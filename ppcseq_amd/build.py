"""Build the HIP C-ABI library (libppcx.so) in-tree with hipcc for gfx950."""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libppcx.so")
# the testing build (-DPPCX_TESTING: fault injection, forced cell paths, kernel-level timing; csrc/ppcx_testing.h) lives
# with the tests, not in the package
TESTING_LIB = os.path.join(os.path.dirname(HERE), "tests", "libppcx_testing.so")
SOURCES = ["ppcx_kernels.hip", "ppcx_capi.hip"]
HEADERS = ["ppcx_math.h", "ppcx_disp.h", "ppcx_model.h", "ppcx_nuts.h", "ppcx_gene.h", "ppcx_kernels.h", "ppcx_testing.h",
           os.path.join("..", "..", "include", "ppcx.h")]


# hipcc (ROCm 7.2) miscompiles the scalar state machine of ppcx_step_kernel when the SLP vectoriser has made <2 x i32> phis of
# adjacent Cmd fields and AMDGPUCodeGenPrepare breaks them up again: the IR stays correct, the ISA loses a copy on one
# predecessor path (Cmd::rng_c3 = 0 on the halving / doubling path of the step-size search). Traced in round 4 (DESIGN.md
# section 3, profiles/r04_miscompile/): of 15 builds with single passes switched off only -fno-slp-vectorize and
# -mllvm -amdgpu-codegenprepare-break-large-phis=false give correct code. Neither changes a fit's time; the second costs
# ppcx_step_kernel nine vector registers and with them its third workgroup per CU (169 instead of 160; 158 without SLP),
# so the library is built without SLP vectorisation (scalar fp64 kernels: there is nothing for it to vectorise usefully).
CODEGEN_FLAGS = ["-fno-slp-vectorize"]


def _stale(lib: str) -> bool:
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def _compile(lib: str, extra, verbose: bool) -> None:
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    procs = []
    for s in SOURCES:                            # the two translation units side by side
        o = lib + "." + s + ".o"
        objs.append(o)
        cmd = [hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-fvisibility=hidden", "-Wno-unused-result"] + CODEGEN_FLAGS + extra + \
              ["-c", "-o", o, os.path.join(CSRC, s)]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        procs.append(subprocess.Popen(cmd))
    for p in procs:
        if p.wait() != 0:
            raise subprocess.CalledProcessError(p.returncode, p.args)
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs)
    for o in objs:
        os.remove(o)


def build(force: bool = False, verbose: bool = False) -> str:
    if force or _stale(LIB):
        _compile(LIB, [], verbose)
    return LIB


def build_testing(force: bool = False, verbose: bool = False) -> str:
    if force or _stale(TESTING_LIB):
        _compile(TESTING_LIB, ["-DPPCX_TESTING"], verbose)
    return TESTING_LIB


def build_all(force: bool = False, verbose: bool = False):
    """The product and the testing build side by side (four hipcc processes: the kernels' translation unit takes minutes)."""
    import threading
    jobs = []
    if force or _stale(LIB):
        jobs.append((LIB, []))
    if force or _stale(TESTING_LIB):
        jobs.append((TESTING_LIB, ["-DPPCX_TESTING"]))
    errs = []

    def run(lib, extra):
        try:
            _compile(lib, extra, verbose)
        except Exception as e:                   # noqa: BLE001 -- reported below
            errs.append(e)
    th = [threading.Thread(target=run, args=j) for j in jobs]
    for t in th:
        t.start()
    for t in th:
        t.join()
    if errs:
        raise errs[0]
    return LIB, TESTING_LIB


if __name__ == "__main__":
    if "--testing" in sys.argv:
        print(*build_all(force="--force" in sys.argv, verbose=True))
    else:
        print(build(force="--force" in sys.argv, verbose=True))
