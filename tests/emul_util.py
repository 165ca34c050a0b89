"""ctypes helpers for the CPU emulation harness (tests/emul/ppcx_emul.cpp)."""
import ctypes as C

import numpy as np


def P(a, t):
    return a.ctypes.data_as(C.POINTER(t))


class EmulCfg(C.Structure):
    _fields_ = [("chains", C.c_int), ("iter", C.c_int), ("warmup", C.c_int), ("seed", C.c_ulonglong),
                ("adapt_delta", C.c_double), ("max_treedepth", C.c_int), ("init_radius", C.c_double),
                ("stepsize0", C.c_double), ("init_buffer", C.c_int), ("term_buffer", C.c_int), ("window", C.c_int),
                ("chain_id_offset", C.c_int)]


def emul_lp(E, counts, X, exposure, K, u, excl=None, lambda_mu_mu=5.612671):
    cnt = np.ascontiguousarray(counts, np.int32)
    G, S = cnt.shape
    X = np.asfortranarray(np.asarray(X, float).reshape(S, -1))
    ex = np.ascontiguousarray(excl if excl is not None else np.zeros(0), np.int32)
    u = np.ascontiguousarray(u, np.float64)
    lp = C.c_double()
    g = np.zeros_like(u)
    E.emul_log_prob_grad(G, S, X.shape[1], int(K), P(cnt, C.c_int32), P(X, C.c_double),
                         P(np.ascontiguousarray(exposure, np.float64), C.c_double), C.c_double(lambda_mu_mu),
                         int(ex.size), P(ex, C.c_int32), P(u, C.c_double), C.byref(lp), P(g, C.c_double))
    return lp.value, g


def emul_fit(E, counts, X, exposure, K, chains, iter, warmup, seed, excl=None, max_treedepth=10, D=None):
    cnt = np.ascontiguousarray(counts, np.int32)
    G, S = cnt.shape
    X = np.asfortranarray(np.asarray(X, float).reshape(S, -1))
    Cc = X.shape[1]
    ex = np.ascontiguousarray(excl if excl is not None else np.zeros(0), np.int32)
    D = 2 * G + K * max(Cc - 1, 1) + 6
    nk = iter - warmup
    cfg = EmulCfg(chains, iter, warmup, seed, 0.8, max_treedepth, 2.0, 1.0, 75, 50, 25, 0)
    out = dict(draws=np.zeros((chains, nk, D)), lp=np.zeros((chains, nk)), stepsize=np.zeros((chains, iter)),
               treedepth=np.zeros((chains, iter), np.int32), n_leapfrog=np.zeros((chains, iter), np.int32),
               divergent=np.zeros((chains, iter), np.int32), accept=np.zeros((chains, iter)))
    rc = E.emul_fit_nuts(G, S, Cc, int(K), P(cnt, C.c_int32), P(X, C.c_double),
                         P(np.ascontiguousarray(exposure, np.float64), C.c_double), C.c_double(5.612671),
                         int(ex.size), P(ex, C.c_int32), C.byref(cfg), P(out["draws"], C.c_double),
                         P(out["lp"], C.c_double), P(out["stepsize"], C.c_double), P(out["treedepth"], C.c_int),
                         P(out["n_leapfrog"], C.c_int), P(out["divergent"], C.c_int), P(out["accept"], C.c_double))
    assert rc == 0, rc
    return out


def emul_fit_pipelined(E, counts, X, exposure, K, chains, iter, warmup, seed, excl=None, max_treedepth=10, spec=True,
                       ls_first_s=False):
    """The two-launch round protocol (ppcx_ls_kernel + ppcx_gene_kernel) emulated with plain loops; also returns the
    number of rounds per chain and how many of them only carried a command (mis-anticipated positions)."""
    cnt = np.ascontiguousarray(counts, np.int32)
    G, S = cnt.shape
    X = np.asfortranarray(np.asarray(X, float).reshape(S, -1))
    Cc = X.shape[1]
    ex = np.ascontiguousarray(excl if excl is not None else np.zeros(0), np.int32)
    D = 2 * G + K * max(Cc - 1, 1) + 6
    nk = iter - warmup
    cfg = EmulCfg(chains, iter, warmup, seed, 0.8, max_treedepth, 2.0, 1.0, 75, 50, 25, 0)
    out = dict(draws=np.zeros((chains, nk, D)), lp=np.zeros((chains, nk)), stepsize=np.zeros((chains, iter)),
               treedepth=np.zeros((chains, iter), np.int32), n_leapfrog=np.zeros((chains, iter), np.int32),
               divergent=np.zeros((chains, iter), np.int32), accept=np.zeros((chains, iter)),
               rounds=np.zeros(chains, np.int64), carried=np.zeros(chains, np.int64))
    rc = E.emul_fit_nuts_pipelined(G, S, Cc, int(K), P(cnt, C.c_int32), P(X, C.c_double),
                                   P(np.ascontiguousarray(exposure, np.float64), C.c_double), C.c_double(5.612671),
                                   int(ex.size), P(ex, C.c_int32), C.byref(cfg), int(bool(spec)), int(bool(ls_first_s)),
                                   P(out["draws"], C.c_double), P(out["lp"], C.c_double), P(out["stepsize"], C.c_double),
                                   P(out["treedepth"], C.c_int), P(out["n_leapfrog"], C.c_int), P(out["divergent"], C.c_int),
                                   P(out["accept"], C.c_double), P(out["rounds"], C.c_long), P(out["carried"], C.c_long))
    assert rc == 0, rc
    return out
