"""Development aid: whole cfg3 fits (8 chains, 150 + 250) of several builds of the library, alternating on one box.
usage: python scripts/gpu_fit_ab.py libA.so libB.so ... ; env ROUNDS (3), TRIM (testing build: trim_slack_permille), NGROUPS (0 = default; not GROUPS, which bash keeps for itself), CHAINS (8)
Each (lib, seed) runs in a child process (a process binds one build)."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, os, time, json
sys.path.insert(0, %r)
from ppcseq_amd import _lib as L
from ppcseq_amd.synth import synth
if os.environ.get("TRIM"):                      # testing build: slack of a chain group's trimmed launch, per mille
    L.testing_set("trim_slack_permille", int(os.environ["TRIM"]))
if os.environ.get("TRIM_EXTRA"):
    L.testing_set("trim_extra_passes", int(os.environ["TRIM_EXTRA"]))
d = synth(20000, 200, seed=20253)
m = L.Model(d["counts"], d["X"], d["exposure"], d["K"])
m.set_rounds(stream_groups=int(os.environ.get("NGROUPS", 0)))
out = []
for seed in [int(s) for s in sys.argv[1:]]:
    t0 = time.perf_counter(); f = m.fit_nuts(chains=int(os.environ.get("CHAINS", 8)), iter=400, warmup=150, seed=seed); dt = time.perf_counter() - t0
    import hashlib
    tm, kt = f.timing(), f.kernel_times(); sha = hashlib.sha1(f.diagnostics()["lp"].tobytes()).hexdigest()[:10]; f.close()
    out.append(dict(seed=seed, wall=round(dt, 3), lp_sha=sha, grads=tm.grad_evals, us_per_round=round(1e6 * tm.seconds / max(kt["launch_triples"], 1), 2),
                    ls_us=round(1e3 * kt["loglik_ms"], 2), gene_us=round(1e3 * kt["close_ms"], 2)))
print(json.dumps(out))
''' % ROOT
libs = sys.argv[1:]
rounds = int(os.environ.get("ROUNDS", 3))
res = {l: [] for l in libs}
for r in range(rounds):
    for l in libs:
        env = dict(os.environ, PPCX_LIB=os.path.abspath(l))
        p = subprocess.run([sys.executable, "-c", CHILD, str(1 + r)], env=env, capture_output=True, text=True)
        if p.returncode != 0:
            print(l, "FAILED", p.stderr[-500:]); continue
        o = json.loads(p.stdout.strip().splitlines()[-1])
        res[l] += o
        print(os.path.basename(l), o, flush=True)
for l in libs:
    w = [x["wall"] for x in res[l]]
    if w:
        print("SUMMARY", os.path.basename(l), "wall mean %.3f min %.3f" % (sum(w) / len(w), min(w)),
              "ls_us mean %.2f gene_us mean %.2f" % (sum(x["ls_us"] for x in res[l]) / len(w), sum(x["gene_us"] for x in res[l]) / len(w)))
