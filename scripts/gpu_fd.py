import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ppcseq_amd import _lib as L
from ppcseq_amd.synth import synth
d = synth(20000, 200, seed=20253); K = d["K"]
m = L.Model(d["counts"], d["X"], d["exposure"], K)
rng = np.random.default_rng(0)
u = rng.uniform(-0.3, 0.3, m.D)
u[3:20003] = d["truth"]["intercept"] + rng.normal(0, 0.05, 20000)
u[3 + 20000 + K:3 + 20000 + K + 20000] = d["truth"]["sigma_raw"]
lp0, g0 = m.log_prob_grad(u)
v = rng.normal(size=m.D); v /= np.linalg.norm(v)
print("lp0", lp0, "g.v", g0 @ v)
prev = None
for h in [8e-4, 4e-4, 2e-4, 1e-4, 5e-5]:
    fd = (m.log_prob_grad(u + h * v)[0] - m.log_prob_grad(u - h * v)[0]) / (2 * h)
    rich = None if prev is None else (4 * fd - prev) / 3
    print(h, fd, fd - g0 @ v, None if rich is None else rich - g0 @ v)
    prev = fd
