"""Pins the oracle's NUTS restatement (Stan defaults) on analytic targets and checks
its adaptation schedule. No Stan draws exist to compare against (no R/Stan in the container)."""
import numpy as np


def test_gaussian_target_recovered(oracle):
    D = 40
    mean = np.linspace(-3, 3, D)
    sd = np.exp(np.linspace(-2, 2, D))
    cfg = oracle.cfg(chains=4, iter=650, warmup=150, seed=7)
    r = oracle.nuts_gauss(mean, sd, cfg)
    x = r.draws.reshape(-1, D)
    # 2000 draws: mean within 5 MC standard errors (ESS >= 400 conservatively), sd within 12 %
    assert np.max(np.abs(x.mean(0) - mean) / sd) < 5 / np.sqrt(400)
    assert np.all(np.abs(x.std(0) / sd - 1) < 0.12)
    assert r.divergent[:, 150:].sum() == 0
    # dual averaging targets adapt_delta = 0.8 during warmup; sampling acceptance ends up near/above it
    assert 0.7 < r.accept[:, 150:].mean() < 0.98


def test_adaptation_schedule(oracle):
    """warmup = 150 => init_buffer 75, one 25-iteration window, term_buffer 50: the metric changes exactly
    once (after iteration 99), where the step size restarts from the init_stepsize heuristic."""
    D = 10
    cfg = oracle.cfg(chains=1, iter=160, warmup=150, seed=3)
    r = oracle.nuts_gauss(np.zeros(D), np.full(D, 0.01), cfg)
    ss = r.stepsize[0]
    # tiny-scale target: before the metric update the step is ~0.01-scale, after it ~1-scale (dual averaging dips to ~0.1
    # now and then: well separated from the 0.004-0.014 of the unit metric)
    assert ss[60:99].max() < 0.02 and ss[101:150].min() > 0.05 and np.median(ss[101:150]) > 0.3
    assert np.all(ss[150:] == ss[150])          # frozen after warmup


def test_model_posterior_sane(oracle):
    from oracle import independent as ind
    d = ind.synth(60, 12, K=6, seed=5)
    m = oracle.model(d["counts"], d["X"], d["exposure"], d["K"])
    r = oracle.nuts_model(m, oracle.cfg(chains=3, iter=300, warmup=150, seed=11))
    x = r.draws.reshape(-1, r.draws.shape[-1])
    assert np.corrcoef(x[:, 3:63].mean(0), d["truth"]["intercept"])[0, 1] > 0.98
    assert r.divergent[:, 150:].sum() == 0


def test_rng_is_counter_based_and_reproducible(oracle):
    a = [oracle.nb2_log_rng(3.0, 2.0, 7, c, d) for c in range(5) for d in range(5)]
    b = [oracle.nb2_log_rng(3.0, 2.0, 7, c, d) for c in range(5) for d in range(5)]
    assert a == b and len(set(a)) > 5
    x = np.array([oracle.nb2_log_rng(np.log(50.0), 4.0, 1, 0, d) for d in range(20000)], dtype=float)
    # NB(mean 50, size 4): var = 50 + 2500/4 = 675
    assert abs(x.mean() - 50) < 1.0 and abs(x.var() - 675) < 60
    y = np.array([oracle.nb2_log_rng(np.log(3.0), 0.5, 1, 1, d) for d in range(20000)], dtype=float)
    assert abs(y.mean() - 3) < 0.15 and abs(y.var() - (3 + 9 / 0.5)) < 3.0


def test_oracle_advi_sits_on_the_nuts_posterior(oracle):
    """The ADVI restatement (rstan::vb, mean-field) converges and its means agree with the NUTS posterior means;
    mean-field standard deviations are narrower, as expected."""
    from oracle import independent as ind
    d = ind.synth(40, 10, K=4, seed=21)
    m = oracle.model(d["counts"], d["X"], d["exposure"], 4, n_threads=4)
    r = oracle.advi(m, output_samples=300, seed=3)
    assert r["converged"] and r["iterations"] % 100 == 0 and r["eta"] in (100, 10, 1, 0.1, 0.01)
    x = oracle.nuts_model(m, oracle.cfg(chains=4, iter=400, warmup=150, seed=3)).draws.reshape(-1, r["mu"].size)
    assert np.corrcoef(r["mu"][3:43], x[:, 3:43].mean(0))[0, 1] > 0.995
    ratio = np.exp(r["omega"][3:43]) / x[:, 3:43].std(0)
    assert 0.6 < np.median(ratio) < 1.15


def test_small_factor_problem_diverges_on_the_oracle_too(oracle):
    """The 90-gene, 12-sample, three-level-factor problem of tests/test_gpu_multi.py::test_do_inference_with_the_genes_sharded_
    over_two_ranks ends with a few divergent transitions after warm-up on the GPU (a RuntimeWarning of do_inference, as rstan
    prints one). The oracle's sampler meets them on the same data and seed: they belong to the problem (twelve samples per gene
    leave the dispersions weakly identified), not to the device path or the sharding."""
    from oracle import independent as ind
    d = ind.synth_factor(90, 12, 7, (3,), 5)
    m = oracle.model(d["counts"], d["X"], d["exposure"], 7, excl=np.array([3, 14], np.int32), n_threads=4)
    n = [int(oracle.nuts_model(m, oracle.cfg(chains=4, iter=450, warmup=150, seed=s)).divergent[:, 150:].sum()) for s in (31, 1, 3)]
    assert all(v <= 12 for v in n) and sum(n) >= 2, n
