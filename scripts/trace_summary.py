"""Summary of a wave trace written by a -DPPCX_TRACE_WAVES build (development aid; see profiles/README.md)."""
import numpy as np, sys
a = np.loadtxt(sys.argv[1])
idx, t0, t1, t2, hw, xcc, n = a.T
T0 = t0.min(); hw = hw.astype(int); xcc = xcc.astype(int) & 0xf
end = t2 - T0
print("wavefronts", len(a), " launch span", (t2.max() - T0) / 100, "us (timestamps in 10 ns ticks of the 100 MHz counter)")
print("wavefront start [us], percentiles 0/50/90/100:", np.percentile(t0 - T0, [0, 50, 90, 100]) / 100)
print("LDS fill done   [us], percentiles 0/50/90/100:", np.percentile(t1 - T0, [0, 50, 90, 100]) / 100)
print("wavefront end   [us], percentiles 0/10/25/50/75/90/100:", np.percentile(end, [0, 10, 25, 50, 75, 90, 100]) / 100)
print("genes per wavefront:", dict(zip(*[x.tolist() for x in np.unique(n, return_counts=True)])))
key = (xcc * 1000 + ((hw >> 13) & 7) * 100 + ((hw >> 12) & 1) * 50 + ((hw >> 8) & 0xf)) * 4 + ((hw >> 4) & 3)
sk = np.unique(key)
s_end = np.array([end[key == k].max() for k in sk]); s_n = np.array([(key == k).sum() for k in sk]); s_g = np.array([n[key == k].sum() for k in sk])
print("SIMDs", len(sk), " wavefronts per SIMD:", dict(zip(*[x.tolist() for x in np.unique(s_n, return_counts=True)])))
print("last wavefront of a SIMD ends [us], percentiles 0/10/50/90/100:", np.percentile(s_end, [0, 10, 50, 90, 100]) / 100)
for g in np.unique(s_g):
    print(f"  SIMDs with {int(g)} genes: {(s_g == g).sum():4d}, end mean {s_end[s_g == g].mean() / 100:.1f} us")
blk = (idx // 4).astype(int)
print("correlation(workgroup id, wavefront end) = %.2f (the oldest wavefront of a SIMD is served first)" % np.corrcoef(blk, end)[0, 1])
