"""Development aid: where a chain's state machine -- the workgroup beside the log-likelihood workgroups in the merged launch of a pipelined
round -- spends its time (testing build: ppcx_testing_sm_trace), for cfg3 fits of 1, 3 and 8 chains, beside the fit's time per round."""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ppcseq_amd import _lib as L, build
L.use_library(os.environ.get("PPCX_LIB") or build.build_testing())
from ppcseq_amd.synth import synth
lib = L.load()
lib.ppcx_testing_sm_trace.argtypes = [C.POINTER(C.c_double)]
d = synth(20000, 200, seed=20253)
m = L.Model(d["counts"], d["X"], d["exposure"], d["K"])
out = (C.c_double * 6)()
for nch, groups in ((1, 0), (3, 0), (8, 1), (8, 0)):
    m.set_rounds(stream_groups=groups)
    lib.ppcx_testing_sm_trace(out)                 # reset
    t0 = time.perf_counter(); f = m.fit_nuts(chains=nch, iter=400, warmup=150, seed=1); dt = time.perf_counter() - t0
    kt = f.kernel_times(); f.close()
    lib.ppcx_testing_sm_trace(out)
    print(f"chains {nch} groups {groups or 'default'}: fit {dt:.3f} s, {1e6 * dt / kt['launch_triples']:.2f} us per round issued; state machine per round: "
          f"loads + slab {out[0]:.2f} us, fold + staging {out[1]:.2f}, exchange {out[2]:.2f}, chain_step {out[3]:.2f}, tail {out[4]:.2f}  (total {sum(out[:5]):.2f}; {int(out[5])} rounds)", flush=True)
