"""Development aid: the pipelined round (two launches) against the three-launch round -- same chains on small models,
then wall time per fit at cfg3 for both, alone and with chain groups on streams."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ppcseq_amd import _lib as L
from ppcseq_amd.synth import synth


def fit(m, pipe, groups=1, **kw):
    """pipe: 0 = three-launch round, 1 = pipelined (two launches)"""
    m.set_rounds(pipelined=0 if pipe == 0 else -1, stream_groups=groups)
    t0 = time.perf_counter()
    f = m.fit_nuts(**kw)
    dt = time.perf_counter() - t0
    return f, dt


ok = True
for (G, S, K, seed) in [(64, 21, 5, 1), (300, 40, 20, 2), (2000, 64, 100, 3)]:
    d = synth(G, S, K=K, seed=seed)
    m = L.Model(d["counts"], d["X"], d["exposure"], K)
    f0, _ = fit(m, 0, chains=3, iter=80, warmup=50, seed=7)
    f1, _ = fit(m, 1, chains=3, iter=80, warmup=50, seed=7)
    d0, d1 = f0.diagnostics(), f1.diagnostics()
    n = 12
    same = np.array_equal(d0["n_leapfrog"][:, :n], d1["n_leapfrog"][:, :n])
    ss = np.max(np.abs(d0["stepsize"][:, :n] - d1["stepsize"][:, :n]))
    tot0, tot1 = d0["n_leapfrog"].sum(), d1["n_leapfrog"].sum()
    k0, k1 = f0.kernel_times(), f1.kernel_times()
    print(f"G={G} S={S}: first {n} tree sizes equal: {same}; stepsize diff {ss:.2e}; leapfrogs {tot0} vs {tot1}; "
          f"rounds {k0['launch_triples']} vs {k1['launch_triples']}; lp mean {d0['lp'].mean():.3f} vs {d1['lp'].mean():.3f}", flush=True)
    ok = ok and same and ss < 1e-8
    f0.close(); f1.close(); m.close()

d = synth(20000, 200, seed=20253)
m = L.Model(d["counts"], d["X"], d["exposure"], d["K"])
res = {}
for rep in range(2):
    for pipe in [int(x) for x in os.environ.get("MODES", "0,1").split(",")]:
        for groups in [int(x) for x in os.environ.get("NGROUPS", "1,2,3").split(",")]:
            f, dt = fit(m, pipe, groups, chains=8, iter=400, warmup=150, seed=1 + rep)
            tm, kt = f.timing(), f.kernel_times()
            dg = f.diagnostics()
            res.setdefault((pipe, groups), []).append(dt)
            print(f"cfg3 pipe={pipe} groups={groups} rep={rep}: wall {dt:.3f} s, pump {tm.seconds:.3f} s, grad evals {tm.grad_evals}, "
                  f"rounds {kt['launch_triples']}, us/round {1e6 * tm.seconds / max(kt['launch_triples'], 1):.1f}, "
                  f"kernels ms {kt['loglik_ms']*1e3:.1f}/{kt['close_ms']*1e3:.1f}/{kt['update_ms']*1e3:.1f} us, "
                  f"div {int(dg['divergent'][:, 150:].sum())}, lp mean {dg['lp'].mean():.1f}", flush=True)
            f.close()
for k, v in sorted(res.items()):
    print("SUMMARY pipe=%d groups=%d: min %.3f s  mean %.3f s" % (k[0], k[1], min(v), sum(v) / len(v)))
m.close()
print("PIPE_CHECK_OK" if ok else "PIPE_CHECK_MISMATCH")
sys.exit(0 if ok else 1)
