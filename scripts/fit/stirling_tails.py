"""Coefficients of the Stirling-tail polynomials of ppcx_math.h (stirling_tails).

    lgamma(x)  = (x - 1/2) ln x - x + ln(2 pi)/2 + r F(r^2)            r = 1/x
    digamma(x) = ln x - r/2 - r^2 G(r^2)

F and G are smooth on r^2 in [0, 1/64] (x >= 8). The asymptotic (Bernoulli) series needs 7 terms for 2e-15 at x = 8;
a polynomial interpolating F, G at Chebyshev nodes of [0, 1/64] (close to the minimax polynomial) reaches 4e-16 with
5 coefficients. This script computes them with mpmath (50 digits), rounds to double and reports the maximum absolute
error of the double-precision Horner evaluation over x in [8, 1e6]. Development aid: its output is pasted into ppcx_math.h.
"""
import mpmath as mp
import numpy as np

mp.mp.dps = 60
N = 5
XMIN = 8


def lgtail(x):
    return mp.loggamma(x) - ((x - mp.mpf(1) / 2) * mp.log(x) - x + mp.log(2 * mp.pi) / 2)


def dgtail(x):
    return mp.log(x) - mp.digamma(x)


def F(t):
    if t == 0:
        return mp.mpf(1) / 12
    x = 1 / mp.sqrt(t)
    return lgtail(x) * x


def G(t):
    if t == 0:
        return mp.mpf(1) / 12
    x = 1 / mp.sqrt(t)
    return (dgtail(x) - 1 / (2 * x)) * x * x


def cheb_fit(f, n, b):
    nodes = [b / 2 * (1 + mp.cos(mp.pi * (2 * k + 1) / (2 * n))) for k in range(n)]
    A = mp.matrix(n, n)
    y = mp.matrix(n, 1)
    for i, t in enumerate(nodes):
        for j in range(n):
            A[i, j] = t ** j
        y[i] = f(t)
    return [mp.lu_solve(A, y)[j] for j in range(n)]


def horner(c, r2):
    t = np.full_like(r2, c[-1])
    for k in range(len(c) - 2, -1, -1):
        t = t * r2 + c[k]          # numpy has no fma: the product's rounding only adds to the reported error
    return t


if __name__ == "__main__":
    b = 1 / mp.mpf(XMIN) ** 2
    cF = [float(c) for c in cheb_fit(F, N, b)]
    cG = [float(c) for c in cheb_fit(G, N, b)]
    xs = np.concatenate([np.linspace(8, 64, 4001), np.geomspace(64, 1e6, 2001)])
    rx = 1.0 / xs
    r2 = rx * rx
    lg = rx * horner(cF, r2)
    dg = 0.5 * rx + r2 * horner(cG, r2)
    e_lg = max(abs(mp.mpf(float(a)) - lgtail(mp.mpf(float(x)))) for a, x in zip(lg, xs))
    e_dg = max(abs(mp.mpf(float(a)) - dgtail(mp.mpf(float(x)))) for a, x in zip(dg, xs))
    print("// lg_tail(r) = r * (F0 + r2 (F1 + r2 (F2 + r2 (F3 + r2 F4)))), max abs error %s on x >= 8" % mp.nstr(e_lg, 3))
    print("constexpr double kStirlingF[5] = {%s};" % ", ".join("%.17e" % c for c in cF))
    print("// dg_tail(r) = r/2 + r2 * (G0 + r2 (G1 + r2 (G2 + r2 (G3 + r2 G4)))), max abs error %s on x >= 8" % mp.nstr(e_dg, 3))
    print("constexpr double kStirlingG[5] = {%s};" % ", ".join("%.17e" % c for c in cG))
