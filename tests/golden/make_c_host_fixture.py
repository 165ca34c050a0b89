"""Writes tests/c_host/bundled_53x21.txt: the inputs of the reference's testthat case (tests/testthat/test-ppcSeq.R:11-24 --
bundled `counts`, SLC16A12 / CYP1A1 / ART3 + the 50 least significant genes, `~ Label`) in the plain-text form the
C host (tests/c_host/dot_c_host.c) reads: `G S C K`, G rows of S counts, the S x C design matrix column by column, the S
exposure rates (TMM, ppcseq_amd.methods.get_scaled_counts_bulk). Data only; run from the repository root."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests.conftest import bundled_test_config  # noqa: E402
from ppcseq_amd.methods import get_scaled_counts_bulk  # noqa: E402

z = np.load(os.path.join(ROOT, "tests", "golden", "counts_bundled.npz"), allow_pickle=False)
bundled = {k: z[k] for k in z.files}
counts, X, genes, K = bundled_test_config(bundled)
samples = [str(s) for s in bundled["samples"]]
mult, _ = get_scaled_counts_bulk(counts, samples)
expo = -np.log(np.array([mult[s] for s in samples]))
G, S = counts.shape
with open(os.path.join(ROOT, "tests", "c_host", "bundled_53x21.txt"), "w") as f:
    f.write(f"{G} {S} {X.shape[1]} {K}\n")
    for g in range(G):
        f.write(" ".join(str(int(v)) for v in counts[g]) + "\n")
    for c in range(X.shape[1]):
        f.write(" ".join(repr(float(v)) for v in X[:, c]) + "\n")
    f.write(" ".join(repr(float(v)) for v in expo) + "\n")
    f.write(" ".join(genes[:K]) + "\n")
print("written", G, S, K)
