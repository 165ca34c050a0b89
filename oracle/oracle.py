"""TEST INFRASTRUCTURE (oracle) -- ctypes front-end to oracle/ppc_oracle.c.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module; the product package (ppcseq_amd) never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


def build(native: bool = False) -> str:
    """Compile the oracle with gcc (idempotent) and return the library path."""
    name = "libppc_oracle_native.so" if native else "libppc_oracle.so"
    path = os.path.join(_HERE, name)
    src = os.path.join(_HERE, "ppc_oracle.c")
    hdr = os.path.join(_HERE, "philox_spec.h")
    if (not os.path.exists(path)) or os.path.getmtime(path) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["make", "-C", _HERE, "native" if native else "all"], stdout=subprocess.DEVNULL)
    return path


class _Model(C.Structure):
    _fields_ = [("G", C.c_int), ("S", C.c_int), ("C", C.c_int), ("K", C.c_int),
                ("counts", C.POINTER(C.c_int32)), ("X", C.POINTER(C.c_double)),
                ("exposure", C.POINTER(C.c_double)), ("lambda_mu_mu", C.c_double),
                ("n_excl", C.c_int), ("excl", C.POINTER(C.c_int32)), ("n_threads", C.c_int)]


class _Cfg(C.Structure):
    _fields_ = [("chains", C.c_int), ("iter", C.c_int), ("warmup", C.c_int), ("seed", C.c_uint64),
                ("adapt_delta", C.c_double), ("max_treedepth", C.c_int), ("init_radius", C.c_double),
                ("stepsize0", C.c_double), ("init_buffer", C.c_int), ("term_buffer", C.c_int),
                ("window", C.c_int), ("max_leapfrogs_total", C.c_int)]


class _AdviCfg(C.Structure):
    _fields_ = [("output_samples", C.c_int), ("iter", C.c_int), ("tol_rel_obj", C.c_double), ("grad_samples", C.c_int),
                ("elbo_samples", C.c_int), ("eval_elbo", C.c_int), ("adapt_iter", C.c_int), ("seed", C.c_uint64),
                ("init_radius", C.c_double)]


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t)) if a is not None else None


@dataclass
class NutsResult:
    draws: np.ndarray        # [chains, n_keep, D] unconstrained
    lp: np.ndarray           # [chains, n_keep]
    stepsize: np.ndarray     # [chains, iter]
    treedepth: np.ndarray
    n_leapfrog: np.ndarray
    divergent: np.ndarray
    accept: np.ndarray
    metric: np.ndarray | None
    iters_done: np.ndarray | None


class Oracle:
    def __init__(self, native: bool = False):
        self.lib = C.CDLL(build(native))
        L = self.lib
        L.ppco_digamma.restype = C.c_double
        L.ppco_digamma.argtypes = [C.c_double]
        L.ppco_dim.restype = C.c_int
        L.ppco_log_prob_grad.restype = C.c_double
        L.ppco_log_prob_grad.argtypes = [C.POINTER(_Model), C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.ppco_nb2_log_rng.restype = C.c_int32
        L.ppco_nb2_log_rng.argtypes = [C.c_double, C.c_double, C.c_uint64, C.c_uint32, C.c_uint32]
        L.ppco_nuts_model.restype = C.c_int
        L.ppco_nuts_gauss.restype = C.c_int
        L.ppco_generated_quantities.restype = None
        L.ppco_summarise.restype = None

    # -- model packing ---------------------------------------------------------------
    def model(self, counts, X, exposure, K, lambda_mu_mu=5.612671, excl=None, n_threads=1):
        counts = np.ascontiguousarray(counts, dtype=np.int32)
        G, S = counts.shape
        X = np.asfortranarray(np.asarray(X, dtype=np.float64).reshape(S, -1))
        exposure = np.ascontiguousarray(exposure, dtype=np.float64)
        excl = np.ascontiguousarray(excl if excl is not None else np.zeros(0), dtype=np.int32)
        m = _Model(G, S, X.shape[1], int(K), _p(counts, C.c_int32), _p(X, C.c_double), _p(exposure, C.c_double),
                   float(lambda_mu_mu), int(excl.size), _p(excl, C.c_int32), int(n_threads))
        m._keep = (counts, X, exposure, excl)
        return m

    def dim(self, G, Cc, K):
        return int(self.lib.ppco_dim(int(G), int(Cc), int(K)))

    def log_prob_grad(self, m, u, want_grad=True):
        u = np.ascontiguousarray(u, dtype=np.float64)
        g = np.empty_like(u) if want_grad else None
        lp = self.lib.ppco_log_prob_grad(C.byref(m), _p(u, C.c_double), _p(g, C.c_double))
        return float(lp), g

    @staticmethod
    def cfg(chains=3, iter=300, warmup=150, seed=1, adapt_delta=0.8, max_treedepth=10, init_radius=2.0,
            stepsize0=1.0, init_buffer=75, term_buffer=50, window=25, max_leapfrogs_total=0):
        return _Cfg(chains, iter, warmup, seed, adapt_delta, max_treedepth, init_radius, stepsize0,
                    init_buffer, term_buffer, window, max_leapfrogs_total)

    def _alloc(self, cfg, D):
        nk = cfg.iter - cfg.warmup
        return dict(draws=np.zeros((cfg.chains, nk, D)), lp=np.zeros((cfg.chains, nk)),
                    stepsize=np.zeros((cfg.chains, cfg.iter)), treedepth=np.zeros((cfg.chains, cfg.iter), np.int32),
                    n_leapfrog=np.zeros((cfg.chains, cfg.iter), np.int32),
                    divergent=np.zeros((cfg.chains, cfg.iter), np.int32), accept=np.zeros((cfg.chains, cfg.iter)))

    def nuts_model(self, m, cfg) -> NutsResult:
        D = self.dim(m.G, m.C, m.K)
        o = self._alloc(cfg, D)
        metric = np.zeros((cfg.chains, D))
        done = np.zeros(cfg.chains, np.int32)
        rc = self.lib.ppco_nuts_model(C.byref(m), C.byref(cfg), _p(o["draws"], C.c_double), _p(o["lp"], C.c_double),
                                      _p(o["stepsize"], C.c_double), _p(o["treedepth"], C.c_int),
                                      _p(o["n_leapfrog"], C.c_int), _p(o["divergent"], C.c_int),
                                      _p(o["accept"], C.c_double), _p(metric, C.c_double), _p(done, C.c_int))
        if rc != 0:
            raise RuntimeError("oracle NUTS: initialisation failed")
        return NutsResult(metric=metric, iters_done=done, **o)

    def nuts_gauss(self, mean, sd, cfg) -> NutsResult:
        mean = np.ascontiguousarray(mean, np.float64)
        sd = np.ascontiguousarray(sd, np.float64)
        D = mean.size
        o = self._alloc(cfg, D)
        rc = self.lib.ppco_nuts_gauss(C.c_int(D), _p(mean, C.c_double), _p(sd, C.c_double), C.byref(cfg),
                                      _p(o["draws"], C.c_double), _p(o["lp"], C.c_double),
                                      _p(o["stepsize"], C.c_double), _p(o["treedepth"], C.c_int),
                                      _p(o["n_leapfrog"], C.c_int), _p(o["divergent"], C.c_int),
                                      _p(o["accept"], C.c_double))
        if rc != 0:
            raise RuntimeError("oracle NUTS: initialisation failed")
        return NutsResult(metric=None, iters_done=None, **o)

    def advi(self, m, output_samples=1000, iter=50000, tol_rel_obj=0.005, elbo_samples=100, eval_elbo=100, adapt_iter=50,
             seed=1, init_radius=2.0):
        D = self.dim(m.G, m.C, m.K)
        cfg = _AdviCfg(output_samples, iter, tol_rel_obj, 1, elbo_samples, eval_elbo, adapt_iter, seed, init_radius)
        draws = np.zeros((output_samples, D)); mu = np.zeros(D); om = np.zeros(D); info = np.zeros(4)
        self.lib.ppco_advi.restype = C.c_int
        rc = self.lib.ppco_advi(C.byref(m), C.byref(cfg), _p(draws, C.c_double), _p(mu, C.c_double), _p(om, C.c_double), _p(info, C.c_double))
        if rc != 0:
            raise RuntimeError(f"oracle ADVI failed ({rc})")
        return dict(draws=draws, mu=mu, omega=om, iterations=int(info[0]), converged=bool(info[1]), elbo=info[2], eta=info[3])

    def nb2_log_rng(self, eta, phi, seed, cell, draw):
        return int(self.lib.ppco_nb2_log_rng(float(eta), float(phi), int(seed), int(cell), int(draw)))

    def generated_quantities(self, m, draws, truncation_compensation=1.0, seed=1):
        draws = np.ascontiguousarray(draws, np.float64).reshape(-1, self.dim(m.G, m.C, m.K))
        out = np.zeros((draws.shape[0], m.K, m.S), np.int32)
        self.lib.ppco_generated_quantities(C.byref(m), _p(draws, C.c_double), C.c_int(draws.shape[0]),
                                           C.c_double(truncation_compensation), C.c_uint64(seed), _p(out, C.c_int32))
        return out

    def generated_quantities_approx(self, m, draws, n_gen, truncation_compensation=1.0, seed=1):
        """Approximated analysis (R/utilities.R:733-784): n_gen predictive draws per checked cell from the resampled posterior."""
        draws = np.ascontiguousarray(draws, np.float64).reshape(-1, self.dim(m.G, m.C, m.K))
        out = np.zeros((int(n_gen), m.K, m.S), np.int32)
        self.lib.ppco_generated_quantities_approx.restype = None
        self.lib.ppco_generated_quantities_approx(C.byref(m), _p(draws, C.c_double), C.c_int(draws.shape[0]), C.c_int(int(n_gen)),
                                                  C.c_double(truncation_compensation), C.c_uint64(seed), _p(out, C.c_int32))
        return out

    def summarise(self, x, p_lo, p_hi):
        x = np.ascontiguousarray(x, np.int32)
        nd = x.shape[0]
        ncell = int(np.prod(x.shape[1:]))
        out = np.zeros((ncell, 4))
        self.lib.ppco_summarise(_p(x, C.c_int32), C.c_int(nd), C.c_int(ncell), C.c_double(p_lo), C.c_double(p_hi),
                                _p(out, C.c_double))
        return out.reshape(x.shape[1:] + (4,))


class CpuFast:
    """BENCH / TEST INFRASTRUCTURE: optimised CPU comparator (oracle/cpu_fast.cpp) -- the product's formulation of the
    log density and gradient compiled for the host (-O3 -march=native, OpenMP over genes). Loaded only by bench.py's
    cpu_baseline leg and by tests/."""

    def __init__(self):
        path = os.path.join(_HERE, "libppc_cpu_fast.so")
        src = os.path.join(_HERE, "cpu_fast.cpp")
        hdr_dir = os.path.join(os.path.dirname(_HERE), "ppcseq_amd", "csrc")
        deps = [src] + [os.path.join(hdr_dir, h) for h in ("ppcx_math.h", "ppcx_model.h", "ppcx_nuts.h", "ppcx_gene.h")]
        if (not os.path.exists(path)) or os.path.getmtime(path) < max(os.path.getmtime(p) for p in deps):
            subprocess.check_call(["make", "-C", _HERE, "fast"], stdout=subprocess.DEVNULL)
        self.lib = C.CDLL(path)
        self.lib.ppcf_model_create.restype = C.c_void_p
        self.lib.ppcf_model_create.argtypes = [C.c_int] * 4 + [C.POINTER(C.c_int32), C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_double]
        self.lib.ppcf_model_destroy.argtypes = [C.c_void_p]
        self.lib.ppcf_dim.argtypes = [C.c_void_p]
        self.lib.ppcf_log_prob_grad.restype = C.c_double
        self.lib.ppcf_log_prob_grad.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int]

    def model(self, counts, X, exposure, K, lambda_mu_mu=5.612671):
        counts = np.ascontiguousarray(counts, dtype=np.int32)
        G, S = counts.shape
        X = np.asfortranarray(np.asarray(X, dtype=np.float64).reshape(S, -1))
        exposure = np.ascontiguousarray(exposure, dtype=np.float64)
        return self.lib.ppcf_model_create(G, S, X.shape[1], int(K), _p(counts, C.c_int32), _p(X, C.c_double),
                                          _p(exposure, C.c_double), float(lambda_mu_mu))

    def log_prob_grad(self, m, u, threads=1, want_grad=True):
        u = np.ascontiguousarray(u, dtype=np.float64)
        g = np.zeros_like(u) if want_grad else None
        lp = self.lib.ppcf_log_prob_grad(m, _p(u, C.c_double), _p(g, C.c_double), int(threads))
        return float(lp), g

    def free(self, m):
        self.lib.ppcf_model_destroy(m)

    def nuts(self, oracle: "Oracle", counts, X, exposure, K, cfg, threads_per_chain=1, chain_id_offset=0) -> NutsResult:
        """A whole CPU fit: the oracle's NUTS driver (ppco_nuts_chain_fn: Stan-default sampler, same Philox streams as the GPU
        fit) on this comparator's gradient, one host thread per chain (as rstan runs chains on `cores` workers,
        R/utilities.R:1500-1501) and `threads_per_chain` OpenMP threads over genes inside a gradient (map_rect shards,
        R/utilities.R:1383-1386,1479). Each chain owns a model: the evaluation keeps scratch vectors in it."""
        import threading
        L = oracle.lib
        fn = C.cast(self.lib.ppcf_lp_callback, C.c_void_p)
        L.ppco_nuts_chain_fn.restype = C.c_int
        L.ppco_nuts_chain_fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(_Cfg), C.c_int] + [C.c_void_p] * 7
        self.lib.ppcf_set_threads.argtypes = [C.c_void_p, C.c_int]
        models = [self.model(counts, X, exposure, K) for _ in range(cfg.chains)]
        D = int(self.lib.ppcf_dim(models[0]))
        o = oracle._alloc(cfg, D)
        done = np.zeros(cfg.chains, np.int32)

        def run(c):
            self.lib.ppcf_set_threads(models[c], int(threads_per_chain))
            done[c] = L.ppco_nuts_chain_fn(fn, models[c], D, C.byref(cfg), int(chain_id_offset + c),
                                           o["draws"][c].ctypes.data, o["lp"][c].ctypes.data, o["stepsize"][c].ctypes.data,
                                           o["treedepth"][c].ctypes.data, o["n_leapfrog"][c].ctypes.data,
                                           o["divergent"][c].ctypes.data, o["accept"][c].ctypes.data)
        th = [threading.Thread(target=run, args=(c,)) for c in range(cfg.chains)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        for m in models:
            self.free(m)
        if (done < 0).any():
            raise RuntimeError("CPU NUTS: initialisation failed")
        return NutsResult(metric=None, iters_done=done, **o)
