"""Development aid: what the chip does during a fit with chain groups on their own streams, from a rocprofv3 kernel trace
(kernel_trace.csv: one row per dispatch with start / end timestamps in ns). Prints, over the span of the pipelined rounds:
the share of time with 0 / 1 / 2 / 3+ log-likelihood launches in flight, with a gene kernel in flight, with nothing in
flight; mean durations per kernel; the mean gap between a group's consecutive launches (per stream)."""
import csv, sys
import numpy as np

rows = list(csv.DictReader(open(sys.argv[1])))
name = np.array([("ls" if "ppcx_ls_kernel" in r["Kernel_Name"] else "gene" if "ppcx_gene_kernel" in r["Kernel_Name"] else "other") for r in rows])
t0 = np.array([int(r["Start_Timestamp"]) for r in rows], np.int64)
t1 = np.array([int(r["End_Timestamp"]) for r in rows], np.int64)
q = np.array([r.get("Queue_Id", "0") for r in rows])
sel = name != "other"
a, b = t0[sel].min(), t1[sel].max()
span = (b - a) * 1e-3
print(f"dispatches: ls {np.sum(name == 'ls')}, gene {np.sum(name == 'gene')}, other {np.sum(name == 'other')}; span of the rounds {span * 1e-3:.1f} ms")
for k in ("ls", "gene"):
    d = (t1 - t0)[name == k] * 1e-3
    print(f"  {k}: mean {d.mean():.1f} us, median {np.median(d):.1f}, p90 {np.percentile(d, 90):.1f}; sum {d.sum() * 1e-3:.1f} ms = {d.sum() / span:.2f} of the span")


def coverage(mask):
    """time with k launches of the masked kind in flight"""
    ev = np.concatenate([np.stack([t0[mask], np.ones(mask.sum(), np.int64)], 1), np.stack([t1[mask], -np.ones(mask.sum(), np.int64)], 1)])
    ev = ev[np.lexsort((ev[:, 1], ev[:, 0]))]
    lvl = np.cumsum(ev[:, 1])[:-1]
    dt = np.diff(ev[:, 0])
    out = {}
    for k in range(0, 5):
        out[k] = dt[(lvl == k) if k < 4 else (lvl >= 4)].sum() * 1e-3 / span
    return out


print("log-likelihood launches in flight (share of the span): ", {k: round(v, 3) for k, v in coverage(name == "ls").items()})
print("gene kernels in flight:                                ", {k: round(v, 3) for k, v in coverage(name == "gene").items()})
print("any kernel in flight:                                  ", {k: round(v, 3) for k, v in coverage(sel).items()})
# gene kernel in flight while no log-likelihood launch is: the exposed part of the gene kernels
ev = []
for i in np.nonzero(sel)[0]:
    ev.append((t0[i], 0 if name[i] == "ls" else 1, 1)); ev.append((t1[i], 0 if name[i] == "ls" else 1, -1))
ev.sort()
n = [0, 0]; last = ev[0][0]; acc = {"ls only": 0, "gene only": 0, "both": 0, "idle": 0}
for t, k, s in ev:
    key = "both" if n[0] and n[1] else "ls only" if n[0] else "gene only" if n[1] else "idle"
    acc[key] += t - last; last = t; n[k] += s
print("share of the span:", {k: round(v * 1e-3 / span, 3) for k, v in acc.items()})
for qq in np.unique(q[sel]):
    m = sel & (q == qq)
    o = np.argsort(t0[m]); s, e, nm = t0[m][o], t1[m][o], name[m][o]
    gap = (s[1:] - e[:-1]) * 1e-3
    g_lg = gap[(nm[:-1] == "ls") & (nm[1:] == "gene")]; g_gl = gap[(nm[:-1] == "gene") & (nm[1:] == "ls")]
    grid = np.array([int(r["Grid_Size_X"]) if "Grid_Size_X" in r else int(r.get("Grid_Size", 0)) for r in rows])[m][o]
    print(f"queue {qq}: ls grid sizes {dict(zip(*[x.tolist() for x in np.unique(grid[nm == 'ls'], return_counts=True)]))}")
    print(f"queue {qq}: {m.sum()} dispatches; gap ls->gene mean {g_lg.mean():.1f} us (median {np.median(g_lg):.1f}), gene->ls mean {g_gl.mean():.1f} us (median {np.median(g_gl):.1f})")
