"""Generates tests/golden/counts_bundled.npz from the reference's bundled example data
(/root/reference/data/counts.rda, documented at man/counts.Rd:8 and README.md:32-45).

The fixture is DATA (the reference's own example input: 21 samples x 18,801 genes), exported so the
tests can run on the GPU box where /root/reference does not exist. Run once in the build container:
    python tests/golden/make_counts_fixture.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from ppcseq_amd.rda import read_rda, data_frame_columns  # noqa: E402

cols = data_frame_columns(read_rda("/root/reference/data/counts.rda")["counts"])
samples = list(dict.fromkeys(cols["sample"]))
genes = list(dict.fromkeys(cols["symbol"]))
si = {s: i for i, s in enumerate(samples)}
gi = {g: i for i, g in enumerate(genes)}
G, S = len(genes), len(samples)
value = np.zeros((G, S), np.int32)
g_idx = np.array([gi[g] for g in cols["symbol"]])
s_idx = np.array([si[s] for s in cols["sample"]])
value[g_idx, s_idx] = cols["value"]
pvalue = np.zeros(G)
pvalue[g_idx] = cols["PValue"]
fdr = np.zeros(G)
fdr[g_idx] = cols["FDR"]
label = np.empty(S, dtype=object)
label[s_idx] = cols["Label"]
assert len(cols["value"]) == G * S
np.savez_compressed(os.path.join(HERE, "counts_bundled.npz"), value=value, genes=np.array(genes), samples=np.array(samples),
                    PValue=pvalue, FDR=fdr, Label=label.astype(str))
print(G, S, value.sum(), os.path.getsize(os.path.join(HERE, "counts_bundled.npz")))
