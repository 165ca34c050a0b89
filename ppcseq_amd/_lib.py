"""ctypes binding of the C ABI in include/ppcx.h (libppcx.so, built by ppcseq_amd/build.py).

There is deliberately no CPU fallback: if the HIP library is missing or no MI355X is visible the
calls raise. The library is loaded from the package directory (in-tree build).
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PPCX_LIB", os.path.join(_HERE, "libppcx.so"))   # PPCX_LIB: development builds

EXPORTS = [
    "ppcx_version", "ppcx_device_count", "ppcx_last_error", "ppcx_model_create", "ppcx_model_set_exclusions",
    "ppcx_model_set_launch", "ppcx_model_get_launch", "ppcx_model_get_plan", "ppcx_model_dim", "ppcx_model_destroy", "ppcx_log_prob_grad",
    "ppcx_nuts_config_default", "ppcx_fit_nuts", "ppcx_fit_info", "ppcx_fit_from_draws", "ppcx_fit_get_draws", "ppcx_fit_get_columns",
    "ppcx_fit_get_diagnostics", "ppcx_fit_get_timing", "ppcx_fit_get_kernel_times", "ppcx_fit_ppc", "ppcx_fit_free", "ppcx_do_inference_C", "ppcx_model_set_rounds", "ppcx_model_get_rounds", "ppcx_model_set_progress",
    "ppcx_model_create_shard", "ppcx_model_create_shard_strided", "ppcx_fit_nuts_shards", "ppcx_comm_unique_id", "ppcx_comm_create", "ppcx_comm_destroy",
    "ppcx_fit_nuts_comm", "ppcx_advi_config_default", "ppcx_fit_advi", "ppcx_fit_advi_info", "ppcx_fit_advi_iterative",
    "ppcx_guard_decision", "ppcx_device_memory", "ppcx_fit_get_ppc_timing",
    "ppcx_xchg_create", "ppcx_xchg_handle", "ppcx_xchg_connect", "ppcx_xchg_connect_local", "ppcx_xchg_set_timeout", "ppcx_xchg_destroy",
    "ppcx_fit_nuts_xchg", "ppcx_fit_get_xchg_timing", "ppcx_fit_get_inv_metric",
]
ABI_VERSION = 400           # include/ppcx.h PPCX_VERSION this binding was written for


class PpcxError(RuntimeError):
    pass


class NutsConfig(C.Structure):
    _fields_ = [("chains", C.c_int), ("iter", C.c_int), ("warmup", C.c_int), ("seed", C.c_ulonglong),
                ("adapt_delta", C.c_double), ("max_treedepth", C.c_int), ("init_radius", C.c_double),
                ("stepsize0", C.c_double), ("init_buffer", C.c_int), ("term_buffer", C.c_int), ("window", C.c_int),
                ("chain_id_offset", C.c_int)]


class AdviConfig(C.Structure):
    _fields_ = [("output_samples", C.c_int), ("iter", C.c_int), ("tol_rel_obj", C.c_double), ("grad_samples", C.c_int),
                ("elbo_samples", C.c_int), ("eval_elbo", C.c_int), ("adapt_iter", C.c_int), ("seed", C.c_ulonglong),
                ("init_radius", C.c_double)]


PROGRESS_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_longlong, C.c_double)
_lib = None


def use_library(path=None):
    """Bind another build of the library from now on (None: back to the product build). For the tests that need the
    testing build (tests/libppcx_testing.so: fault injection, forced cell paths, kernel-level timing); every Model / Fit /
    Comm of the previous library must have been closed."""
    global _lib, LIB_PATH
    _lib = None
    LIB_PATH = path if path else os.environ.get("PPCX_LIB", os.path.join(_HERE, "libppcx.so"))


def load() -> C.CDLL:
    """Load libppcx.so (raises if it has not been built)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PpcxError(f"{LIB_PATH} not found: run `python -m ppcseq_amd.build` (hipcc, gfx950) first; "
                        "there is no CPU fallback")
    lib = C.CDLL(LIB_PATH)
    lib.ppcx_version.restype = C.c_int
    if lib.ppcx_version() != ABI_VERSION:
        raise PpcxError(f"{LIB_PATH} has ABI version {lib.ppcx_version()}, this binding needs {ABI_VERSION}: rebuild it "
                        "(`python -m ppcseq_amd.build --force`)")
    dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int32)
    lib.ppcx_guard_decision.argtypes = [dp, C.c_int]
    lib.ppcx_device_memory.argtypes = [C.c_int, C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong)]
    lib.ppcx_fit_get_ppc_timing.argtypes = [C.c_void_p, dp, C.POINTER(C.c_longlong)]
    lib.ppcx_last_error.restype = C.c_char_p
    lib.ppcx_model_create.argtypes = [C.c_int] * 5 + [ip, dp, dp, C.c_double, C.c_int, ip, C.POINTER(C.c_void_p)]
    lib.ppcx_model_set_exclusions.argtypes = [C.c_void_p, C.c_int, ip]
    lib.ppcx_model_set_launch.argtypes = [C.c_void_p, C.c_int, C.c_int]
    lib.ppcx_model_get_launch.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.ppcx_model_get_plan.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int32), C.c_int]
    lib.ppcx_model_dim.argtypes = [C.c_void_p]
    lib.ppcx_model_destroy.argtypes = [C.c_void_p]
    lib.ppcx_model_destroy.restype = None
    lib.ppcx_log_prob_grad.argtypes = [C.c_void_p, C.c_int, dp, dp, dp]
    lib.ppcx_nuts_config_default.argtypes = [C.POINTER(NutsConfig)]
    lib.ppcx_nuts_config_default.restype = None
    lib.ppcx_fit_nuts.argtypes = [C.c_void_p, C.POINTER(NutsConfig), C.POINTER(C.c_void_p)]
    lib.ppcx_fit_info.argtypes = [C.c_void_p] + [C.POINTER(C.c_int)] * 4
    lib.ppcx_fit_from_draws.argtypes = [C.c_void_p, C.c_int, C.c_int, dp, C.POINTER(C.c_void_p)]
    lib.ppcx_fit_get_draws.argtypes = [C.c_void_p, dp]
    lib.ppcx_fit_get_columns.argtypes = [C.c_void_p, C.c_int, ip, dp]
    lib.ppcx_fit_get_diagnostics.argtypes = [C.c_void_p, dp, dp, ip, ip, ip, dp]
    lib.ppcx_fit_get_timing.argtypes = [C.c_void_p, dp, C.POINTER(C.c_longlong), dp, C.POINTER(C.c_longlong), dp]
    lib.ppcx_fit_get_kernel_times.argtypes = [C.c_void_p, dp, dp, dp, C.POINTER(C.c_longlong)]
    lib.ppcx_fit_get_inv_metric.argtypes = [C.c_void_p, dp]
    lib.ppcx_fit_ppc.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_ulonglong, C.c_int, C.c_int, dp, ip]
    lib.ppcx_fit_free.argtypes = [C.c_void_p]
    lib.ppcx_model_set_rounds.argtypes = [C.c_void_p, C.c_int, C.c_int]
    lib.ppcx_xchg_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
    lib.ppcx_xchg_handle.argtypes = [C.c_void_p, C.c_char_p]
    lib.ppcx_xchg_connect.argtypes = [C.c_void_p, C.c_char_p]
    lib.ppcx_xchg_connect_local.argtypes = [C.POINTER(C.c_void_p), C.c_int]
    lib.ppcx_xchg_set_timeout.argtypes = [C.c_void_p, C.c_double]
    lib.ppcx_xchg_destroy.argtypes = [C.c_void_p]
    lib.ppcx_xchg_destroy.restype = None
    lib.ppcx_fit_nuts_xchg.argtypes = [C.c_void_p, C.POINTER(NutsConfig), C.c_void_p, C.POINTER(C.c_void_p)]
    lib.ppcx_fit_get_xchg_timing.argtypes = [C.c_void_p, dp, C.POINTER(C.c_longlong)]
    lib.ppcx_model_get_rounds.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.ppcx_model_set_progress.argtypes = [C.c_void_p, PROGRESS_FN, C.c_void_p, C.c_double]
    if hasattr(lib, "ppcx_testing_set"):         # the testing build (csrc/ppcx_testing.h)
        lib.ppcx_testing_set.argtypes = [C.c_char_p, C.c_longlong]
        lib.ppcx_testing_set_nccl_provider.argtypes = [C.c_char_p]
        lib.ppcx_testing_bench_kernel.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, dp, C.POINTER(C.c_int)]
    lib.ppcx_fit_free.restype = None
    lib.ppcx_advi_config_default.argtypes = [C.POINTER(AdviConfig)]
    lib.ppcx_advi_config_default.restype = None
    lib.ppcx_fit_advi.argtypes = [C.c_void_p, C.POINTER(AdviConfig), C.POINTER(C.c_void_p)]
    lib.ppcx_fit_advi_iterative.argtypes = [C.c_void_p, C.POINTER(AdviConfig), C.c_int, C.POINTER(C.c_void_p)]
    lib.ppcx_fit_advi_info.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_double)]
    lib.ppcx_model_create_shard.argtypes = [C.c_int] * 7 + [ip, dp, dp, C.c_double, C.c_int, ip, C.POINTER(C.c_void_p)]
    lib.ppcx_model_create_shard_strided.argtypes = [C.c_int] * 8 + [ip, dp, dp, C.c_double, C.c_int, ip, C.POINTER(C.c_void_p)]
    lib.ppcx_fit_nuts_shards.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.POINTER(NutsConfig), C.POINTER(C.c_void_p)]
    lib.ppcx_comm_unique_id.argtypes = [C.c_char_p]
    lib.ppcx_comm_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_char_p, C.POINTER(C.c_void_p)]
    lib.ppcx_comm_destroy.argtypes = [C.c_void_p]
    lib.ppcx_comm_destroy.restype = None
    lib.ppcx_fit_nuts_comm.argtypes = [C.c_void_p, C.POINTER(NutsConfig), C.c_void_p, C.POINTER(C.c_void_p)]
    _lib = lib
    return lib


def _check(rc: int):
    if rc != 0:
        raise PpcxError(f"ppcx error {rc}: {load().ppcx_last_error().decode()}")


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t)) if a is not None else None


def testing_set(key: str, value: int):
    """A test hook of the testing build (csrc/ppcx_testing.h); the product build has none."""
    lib = load()
    if not hasattr(lib, "ppcx_testing_set"):
        raise PpcxError(f"{LIB_PATH} is not the testing build")
    _check(lib.ppcx_testing_set(key.encode(), int(value)))


def testing_set_nccl_provider(path: str):
    lib = load()
    if not hasattr(lib, "ppcx_testing_set_nccl_provider"):
        raise PpcxError(f"{LIB_PATH} is not the testing build")
    _check(lib.ppcx_testing_set_nccl_provider(path.encode() if path else None))


def device_count() -> int:
    return int(load().ppcx_device_count())


def device_memory(device=0):
    """(free, total) bytes of a HIP device."""
    f, t = C.c_ulonglong(), C.c_ulonglong()
    _check(load().ppcx_device_memory(int(device), C.byref(f), C.byref(t)))
    return int(f.value), int(t.value)


def shard_genes(G_total, g0, stride):
    """The genes of a strided shard (ppcx_model_create_shard_strided): g0, g0 + stride, ... below G_total."""
    return list(range(int(g0), int(G_total), int(stride)))


class Model:
    """Device-resident model inputs (Stan data block in logical form, include/ppcx.h)."""

    def __init__(self, counts, X, exposure_rate, K, lambda_mu_mu=5.612671, excl=None, device=0, shard=None):
        """shard = (G_total, K_total, g0, g1): `counts` then holds only genes [g0, g1) of the whole problem and K is
        ignored (the shard's checked genes are those of the first K_total that fall in its range). shard = (G_total, K_total,
        g0, None, stride): the genes g0, g0 + stride, ... of the whole problem -- rank r of N with (r, None, N) is the
        reference's round-robin deal of genes to shards (R/utilities.R:125-136; shard_genes() picks the rows)."""
        lib = load()
        counts = np.ascontiguousarray(counts, dtype=np.int32)
        if counts.ndim != 2:
            raise ValueError("counts must be G x S")
        self.G, self.S = counts.shape
        X = np.asfortranarray(np.asarray(X, dtype=np.float64).reshape(self.S, -1))
        self.C = X.shape[1]
        self.K = int(K)
        exposure_rate = np.ascontiguousarray(exposure_rate, dtype=np.float64)
        if exposure_rate.shape != (self.S,):
            raise ValueError("exposure_rate must have length S")
        excl = np.ascontiguousarray(excl if excl is not None else np.zeros(0), dtype=np.int32)
        self.X, self.exposure_rate = X, exposure_rate
        h = C.c_void_p()
        self.shard = shard
        if shard is None:
            _check(lib.ppcx_model_create(int(device), self.G, self.S, self.C, self.K, _p(counts, C.c_int32),
                                         _p(X, C.c_double), _p(exposure_rate, C.c_double), float(lambda_mu_mu),
                                         int(excl.size), _p(excl, C.c_int32), C.byref(h)))
        else:
            if len(shard) == 5:
                Gt, Kt, g0, stride = int(shard[0]), int(shard[1]), int(shard[2]), int(shard[4])
                if len(shard_genes(Gt, g0, stride)) != self.G:
                    raise ValueError("counts must hold exactly the genes of the shard")
                self.K = len([g for g in shard_genes(Gt, g0, stride) if g < Kt])
                _check(lib.ppcx_model_create_shard_strided(int(device), Gt, self.S, self.C, Kt, g0, stride, self.G, _p(counts, C.c_int32),
                                                           _p(X, C.c_double), _p(exposure_rate, C.c_double), float(lambda_mu_mu),
                                                           int(excl.size), _p(excl, C.c_int32), C.byref(h)))
            else:
                Gt, Kt, g0, g1 = (int(v) for v in shard)
                if g1 - g0 != self.G:
                    raise ValueError("counts must hold exactly the genes of the shard")
                self.K = max(0, min(g1, Kt) - min(g0, Kt))
                _check(lib.ppcx_model_create_shard(int(device), Gt, self.S, self.C, Kt, g0, g1, _p(counts, C.c_int32),
                                                   _p(X, C.c_double), _p(exposure_rate, C.c_double), float(lambda_mu_mu),
                                                   int(excl.size), _p(excl, C.c_int32), C.byref(h)))
        self._h = h
        self.D = int(lib.ppcx_model_dim(h))

    def set_exclusions(self, excl):
        excl = np.ascontiguousarray(excl if excl is not None else np.zeros(0), dtype=np.int32)
        _check(load().ppcx_model_set_exclusions(self._h, int(excl.size), _p(excl, C.c_int32)))

    def set_launch(self, lanes_per_gene=0, workgroups=0):
        """Pin the log-likelihood kernel's lanes per gene (a power of two <= 64) and/or its number of persistent
        workgroups; 0 = automatic. Results depend on lanes_per_gene only (summation order inside a gene)."""
        _check(load().ppcx_model_set_launch(self._h, int(lanes_per_gene), int(workgroups)))

    def get_plan(self, nchains):
        """(lanes per gene, workgroups per chain, bounds) of the log-likelihood launch planned for `nchains` chains:
        wavefront j of a chain walks the gene-order positions bounds[j] .. bounds[j + 1] - 1 (diagnostic). nchains < 0:
        the launch of -nchains chains of one of several chain groups, at the lanes per gene in force."""
        lanes, nb = C.c_int(), C.c_int()
        _check(load().ppcx_model_get_plan(self._h, int(nchains), C.byref(lanes), C.byref(nb), None, 0))
        b = np.zeros(4 * nb.value + 1, np.int32)
        _check(load().ppcx_model_get_plan(self._h, int(nchains), C.byref(lanes), C.byref(nb), _p(b, C.c_int32), int(b.size)))
        return lanes.value, nb.value, b

    def get_launch(self):
        a, b = C.c_int(), C.c_int()
        _check(load().ppcx_model_get_launch(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def log_prob_grad(self, u):
        u = np.ascontiguousarray(u, dtype=np.float64)
        one = u.ndim == 1
        u2 = u.reshape(-1, self.D)
        lp = np.zeros(u2.shape[0])
        g = np.zeros_like(u2)
        _check(load().ppcx_log_prob_grad(self._h, u2.shape[0], _p(u2, C.c_double), _p(lp, C.c_double), _p(g, C.c_double)))
        return (float(lp[0]), g[0]) if one else (lp, g)

    def fit_nuts(self, chains=3, iter=300, warmup=150, seed=1, adapt_delta=0.8, max_treedepth=10, init_radius=2.0,
                 stepsize0=1.0, init_buffer=75, term_buffer=50, window=25, chain_id_offset=0) -> "Fit":
        cfg = NutsConfig(chains, iter, warmup, seed, adapt_delta, max_treedepth, init_radius, stepsize0,
                         init_buffer, term_buffer, window, chain_id_offset)
        h = C.c_void_p()
        _check(load().ppcx_fit_nuts(self._h, C.byref(cfg), C.byref(h)))
        return Fit(self, h)

    def fit_advi(self, output_samples=1000, iter=50000, tol_rel_obj=0.005, elbo_samples=100, eval_elbo=100, adapt_iter=50,
                 seed=1, init_radius=2.0, max_attempts=1) -> "Fit":
        """Mean-field ADVI (rstan::vb); returns a one-chain Fit holding output_samples draws of the approximation.
        max_attempts > 1: the bounded vb_iterative retry (R/utilities.R:246-278), attempt k with seed + k."""
        cfg = AdviConfig(output_samples, iter, tol_rel_obj, 1, elbo_samples, eval_elbo, adapt_iter, seed, init_radius)
        h = C.c_void_p()
        if max_attempts > 1:
            _check(load().ppcx_fit_advi_iterative(self._h, C.byref(cfg), int(max_attempts), C.byref(h)))
        else:
            _check(load().ppcx_fit_advi(self._h, C.byref(cfg), C.byref(h)))
        return Fit(self, h)

    def fit_from_draws(self, draws) -> "Fit":
        """A Fit over draws produced elsewhere ([chains, n_keep, D], unconstrained): pooled chains of other ranks."""
        draws = np.ascontiguousarray(draws, dtype=np.float64)
        if draws.ndim != 3 or draws.shape[2] != self.D:
            raise ValueError("draws must be [chains, n_keep, D]")
        h = C.c_void_p()
        _check(load().ppcx_fit_from_draws(self._h, draws.shape[0], draws.shape[1], _p(draws, C.c_double), C.byref(h)))
        return Fit(self, h)

    def fit_nuts_comm(self, comm: "Comm", **kw) -> "Fit":
        """This process's gene shard of a multi-GPU fit; partial sums all-reduced over RCCL every leapfrog."""
        cfg = _make_cfg(**kw)
        h = C.c_void_p()
        _check(load().ppcx_fit_nuts_comm(self._h, C.byref(cfg), comm._h, C.byref(h)))
        return Fit(self, h)

    def fit_nuts_xchg(self, xchg: "Xchg", **kw) -> "Fit":
        """This rank's gene shard of a multi-GPU fit; the ranks' partial sums are exchanged directly by the state machines."""
        cfg = _make_cfg(**kw)
        h = C.c_void_p()
        _check(load().ppcx_fit_nuts_xchg(self._h, C.byref(cfg), xchg._h, C.byref(h)))
        return Fit(self, h)

    def set_rounds(self, pipelined=-2, stream_groups=-1):
        """Round structure of this model's NUTS fits: pipelined -1 = wherever the model allows it (the library's default), 0 = the
        three-launch round; stream_groups 0 = by the number of chains (the library's default), n = n chain groups on their own
        streams. An argument left out (-2 / -1) leaves that setting as it is."""
        _check(load().ppcx_model_set_rounds(self._h, int(pipelined), int(stream_groups)))

    def set_progress(self, fn=None, every_seconds=1.0):
        """fn(first_chain, chains, chains_done, rounds, seconds) during a NUTS fit of this model (None: off). A true return
        value ends the fit: the fit call raises PpcxError (PPCX_ERR_CANCELLED, -7), the model stays usable."""
        self._progress_cb = PROGRESS_FN((lambda user, c0, n, done, rounds, sec: 1 if fn(c0, n, done, rounds, sec) else 0) if fn else 0)
        _check(load().ppcx_model_set_progress(self._h, self._progress_cb, None, float(every_seconds)))

    def get_rounds(self, nchains=1):
        """(pipelined, stream_groups) a fit of `nchains` chains would run with."""
        a, b = C.c_int(), C.c_int()
        _check(load().ppcx_model_get_rounds(self._h, int(nchains), C.byref(a), C.byref(b)))
        return bool(a.value), b.value

    def bench_kernel(self, which=0, nchains=1, warm_rounds=40, reps=50, n_merge=1):
        """(ms per launch, command type) of one kernel of the three-launch round -- testing build only."""
        lib = load()
        if not hasattr(lib, "ppcx_testing_bench_kernel"):
            raise PpcxError(f"{LIB_PATH} is not the testing build")
        ms, t = C.c_double(), C.c_int()
        _check(lib.ppcx_testing_bench_kernel(self._h, int(which), nchains, warm_rounds, reps, n_merge, C.byref(ms), C.byref(t)))
        return ms.value, t.value

    def close(self):
        if getattr(self, "_h", None):
            load().ppcx_model_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _make_cfg(chains=3, iter=300, warmup=150, seed=1, adapt_delta=0.8, max_treedepth=10, init_radius=2.0,
              stepsize0=1.0, init_buffer=75, term_buffer=50, window=25, chain_id_offset=0):
    return NutsConfig(chains, iter, warmup, seed, adapt_delta, max_treedepth, init_radius, stepsize0,
                      init_buffer, term_buffer, window, chain_id_offset)


def fit_nuts_shards(models, **kw):
    """Gene-sharded fit with every shard in this process (one device): returns one Fit per shard."""
    cfg = _make_cfg(**kw)
    n = len(models)
    hm = (C.c_void_p * n)(*[m._h for m in models])
    hf = (C.c_void_p * n)()
    _check(load().ppcx_fit_nuts_shards(hm, n, C.byref(cfg), hf))
    return [Fit(m, C.c_void_p(h)) for m, h in zip(models, hf)]


class Comm:
    """RCCL communicator of a gene-sharded multi-GPU fit (one rank per GPU)."""

    @staticmethod
    def unique_id() -> bytes:
        buf = C.create_string_buffer(128)
        _check(load().ppcx_comm_unique_id(buf))
        return buf.raw

    def __init__(self, nranks, rank, unique_id: bytes, device=0):
        h = C.c_void_p()
        _check(load().ppcx_comm_create(int(device), int(nranks), int(rank), unique_id, C.byref(h)))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            load().ppcx_comm_destroy(self._h)
            self._h = None


class Xchg:
    """Direct-exchange group of a gene-sharded fit (include/ppcx.h ppcx_xchg_*): one rank per process and GPU, connected
    through IPC handles that the host layer all-gathers; or all ranks in this process (Xchg.local_group)."""

    def __init__(self, nranks, rank, max_chains, device=0):
        h = C.c_void_p()
        _check(load().ppcx_xchg_create(int(device), int(nranks), int(rank), int(max_chains), C.byref(h)))
        self._h, self.nranks, self.rank = h, int(nranks), int(rank)

    def handle(self) -> bytes:
        buf = C.create_string_buffer(64)
        _check(load().ppcx_xchg_handle(self._h, buf))
        return buf.raw

    def connect(self, handles):
        """handles: the ranks' 64-byte handles in rank order."""
        blob = b"".join(handles)
        if len(blob) != 64 * self.nranks:
            raise ValueError("need one 64-byte handle per rank")
        _check(load().ppcx_xchg_connect(self._h, blob))

    @staticmethod
    def local_group(n, max_chains, devices=None):
        xs = [Xchg(n, k, max_chains, device=(devices[k] if devices else 0)) for k in range(n)]
        arr = (C.c_void_p * n)(*[x._h for x in xs])
        _check(load().ppcx_xchg_connect_local(arr, n))
        return xs

    def set_timeout(self, seconds):
        _check(load().ppcx_xchg_set_timeout(self._h, float(seconds)))

    def close(self):
        if getattr(self, "_h", None):
            load().ppcx_xchg_destroy(self._h)
            self._h = None


@dataclass
class Timing:
    seconds: float
    grad_evals: int
    gene_kernel_ms_mean: float
    gene_kernel_samples: int
    gene_kernel_chain_launches_mean: float


class Fit:
    """Kept draws + diagnostics of one NUTS run, resident on the device."""

    def __init__(self, model: Model, handle):
        self.model, self._h = model, handle
        c, k, d, it = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        _check(load().ppcx_fit_info(handle, C.byref(c), C.byref(k), C.byref(d), C.byref(it)))
        self.chains, self.n_keep, self.D, self.iter = c.value, k.value, d.value, it.value

    def draws(self):
        out = np.zeros((self.chains, self.n_keep, self.D))
        _check(load().ppcx_fit_get_draws(self._h, _p(out, C.c_double)))
        return out

    def columns(self, cols):
        cols = np.ascontiguousarray(cols, dtype=np.int32)
        out = np.zeros((self.chains, self.n_keep, cols.size))
        _check(load().ppcx_fit_get_columns(self._h, int(cols.size), _p(cols, C.c_int32), _p(out, C.c_double)))
        return out

    def diagnostics(self):
        lp = np.zeros((self.chains, self.n_keep))
        ss = np.zeros((self.chains, self.iter))
        acc = np.zeros((self.chains, self.iter))
        td = np.zeros((self.chains, self.iter), np.int32)
        nl = np.zeros((self.chains, self.iter), np.int32)
        dv = np.zeros((self.chains, self.iter), np.int32)
        _check(load().ppcx_fit_get_diagnostics(self._h, _p(lp, C.c_double), _p(ss, C.c_double), _p(td, C.c_int32),
                                                _p(nl, C.c_int32), _p(dv, C.c_int32), _p(acc, C.c_double)))
        return dict(lp=lp, stepsize=ss, treedepth=td, n_leapfrog=nl, divergent=dv, accept=acc)

    def inv_metric(self):
        """[chains, D] diagonal of the adapted inverse metric (rstan::get_adaptation_info)."""
        out = np.zeros((self.chains, self.D))
        _check(load().ppcx_fit_get_inv_metric(self._h, _p(out, C.c_double)))
        return out

    def timing(self) -> Timing:
        s, ms, cl = C.c_double(), C.c_double(), C.c_double()
        ge, ns = C.c_longlong(), C.c_longlong()
        _check(load().ppcx_fit_get_timing(self._h, C.byref(s), C.byref(ge), C.byref(ms), C.byref(ns), C.byref(cl)))
        return Timing(s.value, ge.value, ms.value, ns.value, cl.value)

    def advi_info(self):
        it, cv = C.c_int(), C.c_int()
        el, et = C.c_double(), C.c_double()
        _check(load().ppcx_fit_advi_info(self._h, C.byref(it), C.byref(cv), C.byref(el), C.byref(et)))
        return dict(iterations=it.value, converged=bool(cv.value), elbo=el.value, eta=et.value)

    def xchg_timing(self):
        """(mean us a chain's state machine waited for its peers per exchange, exchanges) of a direct-exchange fit."""
        us, n = C.c_double(), C.c_longlong()
        _check(load().ppcx_fit_get_xchg_timing(self._h, C.byref(us), C.byref(n)))
        return us.value, int(n.value)

    def ppc_timing(self):
        """(kernel ms, NB draws) of the last ppc() call on this fit."""
        ms, n = C.c_double(), C.c_longlong()
        _check(load().ppcx_fit_get_ppc_timing(self._h, C.byref(ms), C.byref(n)))
        return ms.value, int(n.value)

    def kernel_times(self):
        a, b, c = C.c_double(), C.c_double(), C.c_double()
        n = C.c_longlong()
        _check(load().ppcx_fit_get_kernel_times(self._h, C.byref(a), C.byref(b), C.byref(c), C.byref(n)))
        return dict(loglik_ms=a.value, close_ms=b.value, update_ms=c.value, launch_triples=n.value)

    def ppc(self, truncation_compensation=1.0, p_lo=0.025, p_hi=0.975, seed=1, n_gen=0, resample=False,
            return_counts_rng=False):
        K, S = self.model.K, self.model.S
        ci = np.zeros((K, S, 4))
        n = n_gen if n_gen > 0 else self.chains * self.n_keep
        rng = np.zeros((n, K, S), np.int32) if return_counts_rng else None
        _check(load().ppcx_fit_ppc(self._h, float(truncation_compensation), float(p_lo), float(p_hi), int(seed),
                                   int(n_gen), int(bool(resample)), _p(ci, C.c_double), _p(rng, C.c_int32)))
        return (ci, rng) if return_counts_rng else ci

    def close(self):
        if getattr(self, "_h", None):
            load().ppcx_fit_free(self._h)
            self._h = None

    # a device-resident handle: copies (pandas deep-copies DataFrame.attrs, where identify_outliers(pass_fit=True) puts the
    # fits, R/methods.R:353-357) share it
    def __copy__(self):
        return self

    def __deepcopy__(self, memo):
        return self

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
