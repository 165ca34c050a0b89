#!/bin/bash
# development aid: SQ counters of one kernel. Default: the loglik kernel under the kernel-level bench (scripts/gpu_kbench.py; G, S,
# LANES from the env). SCRIPT=scripts/gpu_ppc_bench.py KERNEL=ppc_wave: the posterior-predictive kernel at the bench's workload.
# usage: [SCRIPT=...] [KERNEL=...] scripts/gpu_sq_pmc.sh <tag> [passes: 1 2 3]
export TMPDIR=/tmp
TAG=${1:-run}; shift
PASSES=${@:-1 2 3}
OUT=gpurun_out/sqpmc_$TAG; rm -rf $OUT; mkdir -p $OUT
P[1]="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY"
P[2]="SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD"
P[3]="GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_IFETCH SQ_ACTIVE_INST_FLAT SQ_VALU_MFMA_BUSY_CYCLES"
for i in $PASSES; do
  rocprofv3 --pmc ${P[$i]} --kernel-include-regex "${KERNEL_RE:-loglik}" --kernel-trace --output-format csv -d $OUT/p$i -- python3 ${SCRIPT:-scripts/gpu_kbench.py} ${SCRIPT_ARGS} > $OUT/p$i.log 2> $OUT/p$i.err || echo "pass $i failed"
done
python3 - $OUT <<'PY'
import csv, glob, collections, sys, os
out = sys.argv[1]
acc = collections.defaultdict(lambda: [0.0, 0, []])
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if os.environ.get("KERNEL", "loglik") not in row.get("Kernel_Name", ""): continue
        a = acc[row["Counter_Name"]]; a[0] += float(row["Counter_Value"]); a[1] += 1; a[2].append(float(row["Counter_Value"]))
with open(out + "/summary.txt", "w") as o:
    for k, (s, n, v) in sorted(acc.items()):
        v.sort()
        o.write(f"{k:28s} dispatches {n:5d} mean/dispatch {s/max(n,1):16.1f} median {v[len(v)//2] if v else 0.0:16.1f}\n")
print(open(out + "/summary.txt").read())
PY
rm -rf $OUT/p1 $OUT/p2 $OUT/p3
