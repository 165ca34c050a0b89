"""Development aid: one chain's warm-up, iteration by iteration (step size, tree depth, leapfrogs, divergences, lp) -- for two
builds of the library side by side (PPCX_LIB_A, PPCX_LIB_B run in child processes)."""
import os, sys, json, subprocess
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    from ppcseq_amd import _lib as L
    L.use_library(os.environ["PPCX_LIB"])
    from ppcseq_amd.synth import synth
    d = synth(20000, 200, seed=20253)
    m = L.Model(d["counts"], d["X"], d["exposure"], d["K"])
    f = m.fit_nuts(chains=1, iter=int(os.environ.get("ITER", 160)), warmup=150, seed=int(os.environ.get("SEED", 2)), chain_id_offset=int(os.environ.get("CHAIN", 0)))
    dg = f.diagnostics()
    out = {k: np.asarray(dg[k])[0].tolist() for k in ("stepsize", "treedepth", "n_leapfrog", "divergent", "accept")}
    dr = f.draws()[0]
    G = 20000
    out["sigma_raw_range_last"] = [float(dr[-1, 3 + G + d["K"]:3 + 2 * G + d["K"]].min()), float(dr[-1, 3 + G + d["K"]:3 + 2 * G + d["K"]].max())]
    print("JSON" + json.dumps(out))
    sys.exit(0)
res = {}
for tag in ("A", "B"):
    env = dict(os.environ, PPCX_LIB=os.environ["PPCX_LIB_" + tag])
    o = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True)
    line = [l for l in o.stdout.splitlines() if l.startswith("JSON")]
    if not line:
        print(tag, "failed:", o.stderr[-2000:]); sys.exit(1)
    res[tag] = json.loads(line[0][4:])
a, b = res["A"], res["B"]
print("sigma_raw range of the last draw: A", a["sigma_raw_range_last"], "B", b["sigma_raw_range_last"])
print("iter : stepsize A / B : depth A / B : leapfrogs A / B : divergent A / B : accept A / B")
for i in range(len(a["stepsize"])):
    print(f"{i:4d} : {a['stepsize'][i]:.5g} / {b['stepsize'][i]:.5g} : {a['treedepth'][i]} / {b['treedepth'][i]} : {a['n_leapfrog'][i]} / {b['n_leapfrog'][i]} : {a['divergent'][i]} / {b['divergent'][i]} : {a['accept'][i]:.3f} / {b['accept'][i]:.3f}")
