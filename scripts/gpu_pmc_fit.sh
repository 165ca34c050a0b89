#!/bin/bash
# development aid: PMC counters per kernel over a short cfg3 fit (scripts/gpu_fit_short.py; mode from the environment).
# usage: scripts/gpu_pmc_fit.sh <tag> "<counters of pass 1>" ["<counters of pass 2>" ...]   (one rocprofv3 run per pass)
export TMPDIR=/tmp
TAG=$1; shift
OUT=gpurun_out/pmcfit_$TAG; rm -rf $OUT; mkdir -p $OUT
i=0
for P in "$@"; do
  i=$((i+1))
  rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT/p$i -- python3 scripts/gpu_fit_short.py > $OUT/p$i.log 2> $OUT/p$i.err || echo "pass $i failed"
done
python3 - $OUT <<'PY'
import csv, glob, collections, sys
out = sys.argv[1]
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "")
        k = k.split("(")[0].replace("void ppcx::", "").replace("ppcx::", "")
        a = acc[(k, row["Counter_Name"])]; a[0] += float(row["Counter_Value"]); a[1] += 1
with open(out + "/summary.txt", "w") as o:
    for (k, c), (s, n) in sorted(acc.items()):
        if n < 50: continue
        o.write(f"{k:34s} {c:26s} dispatches {n:6d} mean/dispatch {s/max(n,1):16.1f}\n")
print(open(out + "/summary.txt").read())
PY
rm -rf $OUT/p[0-9]
