"""Development aid: phase stamps of the log-likelihood kernel's passes (a -DPPCX_TRACE build of the testing library, PPCX_LIB):
per traced wavefront and pass, cycles from the pass's start to the start of the gene's work, its end (sweep + table), the L-lane
sums and the stores."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ppcseq_amd import _lib as L
L.use_library(os.environ["PPCX_LIB"])
from ppcseq_amd.synth import synth
G, S = int(os.environ.get("G", 20000)), int(os.environ.get("S", 200))
d = synth(G, S, seed=20253)
m = L.Model(d["counts"], d["X"], d["exposure"], d["K"])
chains, lanes = int(os.environ.get("CHAINS", 8)), int(os.environ.get("LANES", 8))
m.set_launch(lanes, 0)
path = os.environ.get("PPCX_TRACE_FILE", "gpurun_out/pass_trace.bin")
os.environ["PPCX_TRACE_FILE"] = path
ms, t = m.bench_kernel(0, chains, int(os.environ.get("WARM", 3000)), 20, 1)
tr = np.fromfile(path, dtype=np.uint64).reshape(64, 4, 8, 8).astype(np.int64)
print(f"launch {1e3 * ms:.1f} us")
ok = tr[..., 0] > 0
t0 = tr[..., 0][ok].min()
for name, a, b in (("pass start -> gene work", 0, 1), ("gene work (sweep + table)", 1, 2), ("  before the sweep", 1, 5), ("  the sweep", 5, 6), ("  after the sweep", 6, 2), ("L-lane sums", 2, 3), ("stores", 3, 4), ("whole pass", 0, 4)):
    dlt = (tr[..., b] - tr[..., a])[ok & (tr[..., b] > 0)]
    print(f"{name:28s} cycles: median {np.median(dlt):9.0f}  mean {dlt.mean():9.0f}  p10 {np.percentile(dlt, 10):9.0f}  p90 {np.percentile(dlt, 90):9.0f}")
first = tr[:, :, 0, 0][tr[:, :, 0, 0] > 0]
print("first pass starts (cycles after the earliest): median", np.median(first - t0), "max", (first - t0).max())
last = tr[..., 4].max(axis=2)
print("last stamp (cycles after the earliest start): median", np.median(last[last > 0] - t0), "max", (last[last > 0] - t0).max())
# the clock the chip holds inside the kernel: shader cycles per 100 MHz tick between a wavefront's first and last pass start
dc = tr[:, :, :, 0].max(axis=2) - np.where(ok, tr[:, :, :, 0], np.iinfo(np.int64).max).min(axis=2)
dr = tr[:, :, :, 7].max(axis=2) - np.where(ok, tr[:, :, :, 7], np.iinfo(np.int64).max).min(axis=2)
good = dr > 0
print("in-kernel clock (GHz): median", np.median(dc[good] / dr[good]) * 0.1, "p10", np.percentile(dc[good] / dr[good], 10) * 0.1, "p90", np.percentile(dc[good] / dr[good], 90) * 0.1)
npass = ok.sum(axis=2)
print("passes per traced wavefront:", np.bincount(npass.ravel()))
for w in range(2):
    print("wave", w, "of block 0: per-pass [start, work, end, sums, stores] relative:", (tr[0, w][ok[0, w]] - t0)[:, :5].tolist())
# when the traced workgroups start and end on the 100 MHz clock every XCD shares (s_memtime differs by XCD): the SIMD serves its
# oldest wavefront first, so the quarters of the launch's workgroups finish one after the other
rt = tr[..., 7]
first_rt = np.where(ok, rt, np.iinfo(np.int64).max).min(axis=(1, 2))
t0r = first_rt.min()
clk = np.median(dc[good] / dr[good]) * 100.0      # cycles per us
ends = []
for b in range(tr.shape[0]):
    e = 0.0
    for w in range(4):
        k = npass[b, w] - 1
        if k >= 0:
            e = max(e, (rt[b, w, k] - t0r) * 0.01 + (tr[b, w, k, 4] - tr[b, w, k, 0]) / clk)
    ends.append(e)
ends = np.array(ends)
print("first pass starts, us after the earliest, by quarter of the traced workgroups:", [round(float(np.median((first_rt - t0r)[q * 16:(q + 1) * 16]) * 0.01), 1) for q in range(4)])
print("last pass ends, us after the earliest start, by quarter of the traced workgroups:", [round(float(np.median(ends[q * 16:(q + 1) * 16])), 1) for q in range(4)], "max", round(float(ends.max()), 1))
