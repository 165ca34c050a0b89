"""The optimised CPU comparator of bench.py's cpu_baseline leg (oracle/cpu_fast.cpp: the product's formulation of the
density compiled for the host, OpenMP over genes) against the literal port (oracle/ppc_oracle.c) -- a comparator that
disagrees with the port would time something else. Shapes with every kind of gene: counts below 8 in genes whose other
counts are large (no tail tier for those: ppcx_math.h gene_tier), genes without any small count (tiers 1 and 2), genes
with slopes."""
import numpy as np
import pytest

from oracle import independent as ind
from oracle.oracle import CpuFast, Oracle


@pytest.mark.parametrize("G,S,K,seed", [(40, 30, 6, 1), (25, 64, 0, 2), (12, 7, 12, 3)])
def test_optimised_comparator_agrees_with_the_port(G, S, K, seed):
    d = ind.synth(G, S, K=K, seed=seed)
    counts = d["counts"].copy()
    rng = np.random.default_rng(seed)
    counts[: G // 3] = rng.integers(300, 5000, (G // 3, S))            # tier 2: every count >= 256 ...
    counts[G // 3: 2 * G // 3] = rng.integers(32, 250, (2 * G // 3 - G // 3, S))   # ... tier 1: every count >= 32
    counts[0, 3] = 0; counts[1, 0] = 7; counts[G // 3, 2] = 5          # ... except here: those genes have list cells
    O, F = Oracle(), CpuFast()
    mo = O.model(counts, d["X"], d["exposure"], K)
    mf = F.model(counts, d["X"], d["exposure"], K)
    try:
        D = O.dim(G, d["X"].shape[1], K)
        for k in range(3):
            u = rng.uniform(-0.6, 0.6, D); u[3:3 + G] += 5.0
            lp_o, g_o = O.log_prob_grad(mo, u)
            for threads in (1, 3):
                lp_f, g_f = F.log_prob_grad(mf, u, threads=threads)
                assert abs(lp_f - lp_o) <= 1e-11 * max(1.0, abs(lp_o)), (k, threads)
                assert np.max(np.abs(g_f - g_o) / (1 + np.abs(g_o))) <= 1e-10, (k, threads)
    finally:
        F.free(mf)


def test_whole_cpu_fit_on_the_comparator_follows_the_port():
    """bench.py --cpu-full-cfg2 times a WHOLE CPU fit: the oracle's NUTS driver on the comparator's gradient, one host
    thread per chain (CpuFast.nuts). With gradients that agree to 1e-12 it must take the port's decisions for the first
    iterations (tree sizes, step sizes) before rounding separates the two, whatever the threads per chain."""
    d = ind.synth(40, 12, K=5, seed=3)
    O, F = Oracle(), CpuFast()
    cfg = O.cfg(chains=3, iter=30, warmup=20, seed=4)
    ref = O.nuts_model(O.model(d["counts"], d["X"], d["exposure"], 5), cfg)
    for threads in (1, 2):
        r = F.nuts(O, d["counts"], d["X"], d["exposure"], 5, cfg, threads_per_chain=threads)
        assert np.array_equal(r.n_leapfrog[:, :8], ref.n_leapfrog[:, :8]), threads
        assert np.allclose(r.stepsize[:, :8], ref.stepsize[:, :8], rtol=1e-8, atol=0)
        assert (r.iters_done == 30).all() and np.isfinite(r.draws).all()
