"""Host-side mirror of ppcseq's inference driver `do_inference()` (R/utilities.R:1321-1547).

Same names, argument meaning and error behaviour as the reference for this path; the only thing
replaced is the sampler call: `rstan::sampling(stanmodels$negBinomial_MPI, ...)`
(R/utilities.R:1497-1512) becomes the C-ABI library `libppcx.so` (include/ppcx.h) driving HIP kernels
on the MI355X. The shard packing of the reference (`format_for_MPI` R/utilities.R:125-174,
`counts_package` :1452-1461) is a CPU-threading artefact and is not reproduced: the library takes the
logical G x S matrix. There is no CPU fallback.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np

from . import _lib


def find_optimal_number_of_chains(how_many_posterior_draws, max_number_to_check=100):
    """R/utilities.R:291-303: argmin over c in 2..max of draws/c + 150 c (first minimiser)."""
    best, best_tot = 2, float("inf")
    for c in range(2, max_number_to_check + 1):
        tot = how_many_posterior_draws / c + 150 * c
        if tot < best_tot:
            best, best_tot = c, tot
    return best


def quantile7(x, p):
    """R `quantile(type = 7)` (rstan::summary / stats::quantile default)."""
    x = np.sort(np.asarray(x, dtype=np.float64))
    h = (x.size - 1) * p
    lo = int(math.floor(h))
    if lo >= x.size - 1:
        return float(x[-1])
    return float(x[lo] + (h - lo) * (x[lo + 1] - x[lo]))


@dataclass
class InferenceResult:
    """What `do_inference` returns, as arrays over the checked cells (g < K, all samples).

    Columns of the reference's tibble (R/utilities.R:1516-1544): S, G (1-based), mean, sd, .lower,
    .upper, ppc, `is higher than mean`, slope, `is group high`, deleterious_outliers.
    """
    K: int
    S: int
    mean: np.ndarray            # [K, S]
    sd: np.ndarray
    lower: np.ndarray
    upper: np.ndarray
    ppc: np.ndarray             # bool [K, S]
    is_higher_than_mean: np.ndarray
    slope: np.ndarray           # [K] posterior mean of alpha_sub_1
    is_group_high: np.ndarray | None
    deleterious_outliers: np.ndarray | None
    total_draws: int
    chains: int
    iter: int
    fit: object = None          # the device-resident Fit when pass_fit
    diagnostics: dict = field(default_factory=dict)
    counts_rng: np.ndarray | None = None

    def to_frame(self):
        import pandas as pd
        K, S = self.K, self.S
        g, s = np.meshgrid(np.arange(1, K + 1), np.arange(1, S + 1), indexing="ij")
        d = {"S": s.ravel(), "G": g.ravel(), "mean": self.mean.ravel(), "sd": self.sd.ravel(),
             ".lower": self.lower.ravel(), ".upper": self.upper.ravel(), "ppc": self.ppc.ravel(),
             "is higher than mean": self.is_higher_than_mean.ravel(), "slope": np.repeat(self.slope, S)}
        if self.deleterious_outliers is not None:
            d["is group high"] = self.is_group_high.ravel()
            d["deleterious_outliers"] = self.deleterious_outliers.ravel()
        return pd.DataFrame(d)


def hmc_warnings(diagnostics, warmup, max_treedepth=10):
    """What rstan::sampling tells the reference's user after a NUTS fit (check_hmc_diagnostics: divergent transitions after
    warm-up, transitions that hit the maximum tree depth), as a list of messages. A chain that ends the short warm-up of the
    reference (150 iterations, R/utilities.R:1503) with a step size far below the others' runs every transition at the
    maximum tree depth -- the fit then takes several times as long and this is how the caller learns why."""
    msgs = []
    div = np.asarray(diagnostics["divergent"])[:, warmup:]
    depth = np.asarray(diagnostics["treedepth"])[:, warmup:]
    if div.size and div.sum() > 0:
        msgs.append(f"There were {int(div.sum())} divergent transitions after warmup.")
    n_max = int((depth >= max_treedepth).sum()) if depth.size else 0
    if n_max > 0:
        chains = np.nonzero((depth >= max_treedepth).any(axis=1))[0].tolist()
        msgs.append(f"There were {n_max} transitions after warmup that exceeded the maximum treedepth of {max_treedepth} "
                    f"(chains {chains}): those chains adapted a very small step size and dominate the run time.")
    return msgs


def do_inference(counts, X, exposure_rate, how_many_to_check, *,
                 approximate_posterior_inference=False,
                 approximate_posterior_analysis=False,
                 lambda_mu_mu=5.612671,
                 cores=4,
                 adj_prob_theshold=0.05,
                 how_many_posterior_draws=1000,
                 to_exclude=None,
                 truncation_compensation=1.0,
                 save_generated_quantities=False,
                 pass_fit=False,
                 seed=1,
                 device=0,
                 model=None,
                 chains=None,
                 devices=None,
                 launch=None):
    """One inference pass (discovery or test) of ppcseq on the GPU.

    counts            G x S integer matrix, genes ordered with the `how_many_to_check` checked genes first
                      (R/utilities.R:949-952) and samples in S order (:955-958)
    X                 S x C design matrix (R/utilities.R:887-900)
    exposure_rate     length S, `-log(multiplier)` (R/methods.R:236)
    to_exclude        iterable of (S, G) 1-based pairs as in the reference's tibble (R/methods.R:292-300),
                      or an int array of 0-based cell ids g*S+s
    devices           several HIP devices of this process: the chains are split over them (the reference runs its chains in
                      `cores` worker processes, R/utilities.R:1500-1501) and the credible intervals come from the POOLED
                      draws, as rstan::summary does over merged chains (:685-703). One process per GPU under
                      torch.distributed: ppcseq_amd.distributed.do_inference.
    launch            (lanes_per_gene, workgroups) pins the log-likelihood launch (0 = automatic); by default it follows the number of
                      chains per launch, and results agree to rounding, not bit for bit, between geometries
    Returns an InferenceResult.
    """
    counts = np.asarray(counts)
    if counts.ndim != 2:
        # R/utilities.R:1360-1361
        raise ValueError("The input data frame does not represent a rectangular structure. "
                         "Each transcript must be present in all samples.")
    if not np.issubdtype(counts.dtype, np.integer):
        raise TypeError("The abundance column must be of class integer")   # R/methods.R:146-153
    G, S = counts.shape
    K = int(how_many_to_check)
    X = np.asarray(X, dtype=np.float64).reshape(S, -1)

    # R/utilities.R:1372-1386
    draws_practical = 1000 if approximate_posterior_analysis else how_many_posterior_draws
    if chains is None:
        chains = max(3, min(int(cores), find_optimal_number_of_chains(draws_practical)))
    n_iter = int(math.ceil(draws_practical / chains)) + 150       # R/utilities.R:1502
    warmup = 150                                                    # R/utilities.R:1503

    excl = _to_cell_ids(to_exclude, S)
    if devices is not None and len(devices) > 1 and (save_generated_quantities or pass_fit or model is not None or approximate_posterior_inference):
        raise ValueError("devices=[...] splits the chains of a NUTS fit over several devices and pools their draws: it cannot "
                         "be combined with save_generated_quantities, pass_fit, a caller's model or approximate_posterior_inference")
    if devices is not None and len(devices) > 1 and not approximate_posterior_inference:
        return _do_inference_devices(counts, X, exposure_rate, K, list(devices), chains, n_iter, warmup, excl,
                                     lambda_mu_mu, approximate_posterior_analysis, adj_prob_theshold,
                                     how_many_posterior_draws, truncation_compensation, seed, launch)
    if devices is not None and len(devices) >= 1 and model is None:
        device = devices[0]
    own_model = model is None
    if own_model:
        model = _lib.Model(counts, X, exposure_rate, K, lambda_mu_mu=lambda_mu_mu, excl=excl, device=device)
    else:
        model.set_exclusions(excl)
    if launch is not None:
        model.set_launch(*launch)
    if approximate_posterior_inference:
        # vb_iterative(model, output_samples = draws_practical, iter = 50000, tol_rel_obj = 0.005)
        # (R/utilities.R:1487-1494; the reference passes no seed to vb -- here the run is seeded and reproducible)
        # vb_iterative retries a failed vb() call (R/utilities.R:246-278): here bounded, attempt k with seed + k
        fit = model.fit_advi(output_samples=int(draws_practical), iter=50000, tol_rel_obj=0.005, seed=seed, max_attempts=5)
    else:
        fit = model.fit_nuts(chains=chains, iter=n_iter, warmup=warmup, seed=seed)
    try:
        p = float(adj_prob_theshold)
        if approximate_posterior_analysis:
            # R/utilities.R:733-784: resample the posterior, rnbinom per cell
            out = fit.ppc(truncation_compensation, p, 1 - p, seed=seed, n_gen=int(how_many_posterior_draws),
                          resample=True, return_counts_rng=False)
            ci, rng = out, None
        else:
            out = fit.ppc(truncation_compensation, p, 1 - p, seed=seed, n_gen=0, resample=False,
                          return_counts_rng=bool(save_generated_quantities))
            ci, rng = out if save_generated_quantities else (out, None)
        # slope = posterior mean of alpha_sub_1 (R/utilities.R:1531, :1250-1263)
        off_alpha1 = 3 + G
        slope = fit.columns(np.arange(off_alpha1, off_alpha1 + K)).reshape(-1, K).mean(axis=0) if K else np.zeros(0)
        res = _post_process(counts[:K], ci, slope, X)
        res.total_draws = S * K * int(how_many_posterior_draws)   # R/utilities.R:1544
        res.chains, res.iter = chains, n_iter
        res.diagnostics = fit.advi_info() if approximate_posterior_inference else fit.diagnostics()
        if not approximate_posterior_inference:
            import warnings
            for msg in hmc_warnings(res.diagnostics, warmup):
                warnings.warn(msg, RuntimeWarning, stacklevel=2)
        res.counts_rng = rng
        if pass_fit:
            res.fit = fit
    finally:
        if not pass_fit:
            fit.close()
            if own_model:
                model.close()
    return res


def _to_cell_ids(to_exclude, S):
    if to_exclude is None:
        return np.zeros(0, np.int32)
    a = np.asarray(to_exclude)
    if a.size == 0:
        return np.zeros(0, np.int32)
    if a.ndim == 2 and a.shape[1] == 2:          # (S, G) pairs, 1-based
        return ((a[:, 1].astype(np.int64) - 1) * S + (a[:, 0].astype(np.int64) - 1)).astype(np.int32)
    return a.astype(np.int32).ravel()


def _post_process(counts_checked, ci, slope, X):
    """check_if_within_posterior (R/utilities.R:651-663) + add_deleterious_if_covariate_exists (:493-513)."""
    K, S = counts_checked.shape
    mean, sd, lower, upper = ci[..., 0], ci[..., 1], ci[..., 2], ci[..., 3]
    y = counts_checked.astype(np.float64)
    ppc = (y >= lower) & (y <= upper)                       # dplyr::between is inclusive
    higher = (~ppc) & (y > mean)
    is_group_high = delet = None
    if X.shape[1] > 1:
        f = X[:, 1]
        right = f > f.mean()
        is_group_high = ((slope[:, None] > 0) & right[None, :]) | ((slope[:, None] < 0) & ~right[None, :])
        delet = (~ppc) & (higher == is_group_high)
    return InferenceResult(K=K, S=S, mean=mean, sd=sd, lower=lower, upper=upper, ppc=ppc,
                           is_higher_than_mean=higher, slope=slope, is_group_high=is_group_high,
                           deleterious_outliers=delet, total_draws=0, chains=0, iter=0)


def checked_columns(G, C, K):
    """Columns of the unconstrained vector that belong to the K checked genes and the hyper-parameters, in the order of
    the unconstrained vector of a model that holds ONLY those K genes (Stan declaration order, .stan:183-197)."""
    n2 = max(C - 2, 0)
    off_a1, off_a2 = 3 + G, 3 + G + K
    off_sr = off_a2 + n2 * K
    return np.concatenate([np.arange(3), 3 + np.arange(K), off_a1 + np.arange(K), off_a2 + np.arange(n2 * K),
                           off_sr + np.arange(K), off_sr + G + np.arange(3)]).astype(np.int32)


def pooled_summary(counts, X, exposure_rate, K, draws_checked, *, lambda_mu_mu, approximate_posterior_analysis,
                   adj_prob_theshold, how_many_posterior_draws, truncation_compensation, seed, device=0):
    """Credible intervals, slopes and flags from the pooled draws of all chains (rstan::summary over merged chains,
    R/utilities.R:685-703): `draws_checked` is [chains, n_keep, len(checked_columns)] in global chain order. The
    posterior-predictive kernel runs on a model that holds the K checked genes only -- cell ids g*S+s and draw indices are
    those of the full model, so the result is what a single fit of all the chains gives, bit for bit."""
    counts = np.asarray(counts)
    X = np.asarray(X, dtype=np.float64).reshape(counts.shape[1], -1)
    small = _lib.Model(counts[:K], X, exposure_rate, K, lambda_mu_mu=lambda_mu_mu, device=device)
    try:
        fit = small.fit_from_draws(draws_checked)
        try:
            p = float(adj_prob_theshold)
            if approximate_posterior_analysis:
                ci = fit.ppc(truncation_compensation, p, 1 - p, seed=seed, n_gen=int(how_many_posterior_draws), resample=True)
            else:
                ci = fit.ppc(truncation_compensation, p, 1 - p, seed=seed, n_gen=0, resample=False)
            slope = fit.columns(np.arange(3 + K, 3 + 2 * K)).reshape(-1, K).mean(axis=0) if K else np.zeros(0)
        finally:
            fit.close()
    finally:
        small.close()
    res = _post_process(counts[:K], ci, slope, X)
    res.total_draws = counts.shape[1] * K * int(how_many_posterior_draws)
    return res


def _do_inference_devices(counts, X, exposure_rate, K, devices, chains, n_iter, warmup, excl, lambda_mu_mu,
                          approximate_posterior_analysis, adj_prob_theshold, how_many_posterior_draws,
                          truncation_compensation, seed, launch=None):
    """Chains split over several devices of this process (one host thread per device; the C ABI allows different handles
    on different threads), pooled summary on the first device."""
    import threading
    G, S = counts.shape
    nd = min(len(devices), chains)
    per = int(math.ceil(chains / nd))
    cols = checked_columns(G, X.shape[1], K)
    parts, errs = [None] * nd, [None] * nd

    def work(r):
        n = min(per, chains - r * per)
        if n <= 0:
            return
        try:
            m = _lib.Model(counts, X, exposure_rate, K, lambda_mu_mu=lambda_mu_mu, excl=excl, device=devices[r])
            try:
                if launch is not None:
                    m.set_launch(*launch)
                f = m.fit_nuts(chains=n, iter=n_iter, warmup=warmup, seed=seed, chain_id_offset=r * per)
                try:
                    parts[r] = f.columns(cols)
                finally:
                    f.close()
            finally:
                m.close()
        except Exception as e:          # re-raised on the calling thread
            errs[r] = e

    th = [threading.Thread(target=work, args=(r,)) for r in range(nd)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for e in errs:
        if e is not None:
            raise e
    pooled = np.concatenate([p for p in parts if p is not None], axis=0)
    res = pooled_summary(counts, X, exposure_rate, K, pooled, lambda_mu_mu=lambda_mu_mu,
                         approximate_posterior_analysis=approximate_posterior_analysis, adj_prob_theshold=adj_prob_theshold,
                         how_many_posterior_draws=how_many_posterior_draws, truncation_compensation=truncation_compensation,
                         seed=seed, device=devices[0])
    res.chains, res.iter = chains, n_iter
    return res
