"""Summary of a gene-kernel phase trace written by a -DPPCX_TRACE_GENE build (development aid)."""
import sys
import numpy as np
a = np.loadtxt(sys.argv[1], dtype=np.int64)
nchains = int(sys.argv[2]) if len(sys.argv) > 2 else 8
names = ["start", "loads+barrier", "update", "close math", "reductions/dots", "spec consts", "slab"]
for r in np.unique(a[:, 0]):
    x = a[a[:, 0] == r]
    nb = len(x) // nchains
    t = x[:, 2:9].astype(float)
    ok = t[:, 0] > 0
    if not ok.any():
        continue
    T0 = t[ok, 0].min()
    print(f"round {r}: workgroups traced {ok.sum()}, span first start -> last end {(t[ok].max() - T0) / 100:.2f} us")
    for c in range(nchains):
        xc = x[c * nb:(c + 1) * nb]; tc = xc[:, 2:9].astype(float); okc = tc[:, 0] > 0
        if not okc.any():
            print(f"  chain {c}: not traced (returned early)"); continue
        meta = int(xc[okc][0, 9]); typ, nm, upd, comp = meta >> 32, (meta >> 8) & 0xff, meta & 1, (meta >> 1) & 1
        last = np.array([tc[okc][:, k][tc[okc][:, k] > 0].max() if (tc[okc][:, k] > 0).any() else np.nan for k in range(7)])
        first = tc[okc, 0].min()
        seg = " ".join(f"{names[k]}<={(last[k] - T0) / 100:5.2f}" for k in range(7) if not np.isnan(last[k]))
        print(f"  chain {c}: type {typ} n_merge {nm} update {upd} complete {comp}: first start {(first - T0) / 100:5.2f} | {seg}")
        d = np.diff(tc[okc], axis=1) / 100
        d[tc[okc][:, 1:] <= 0] = np.nan
        with np.errstate(all="ignore"):
            print("           median per-workgroup phase us:", " ".join(f"{names[k + 1]} {np.nanmedian(d[:, k]):.2f}" for k in range(6) if not np.all(np.isnan(d[:, k]))))
