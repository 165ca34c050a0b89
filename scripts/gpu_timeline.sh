#!/bin/bash
# development aid: rocprofv3 kernel trace of a short cfg3 fit (8 chains) and its timeline summary (scripts/timeline_summary.py)
# usage: [PPCX_STREAM_GROUPS=n] scripts/gpu_timeline.sh <tag>
export TMPDIR=/tmp
TAG=${1:-run}
OUT=gpurun_out/timeline_$TAG; rm -rf $OUT; mkdir -p $OUT
ITER=${ITER:-70} WARMUP=${WARMUP:-40} rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 scripts/gpu_fit_short.py > $OUT/fit.log 2> $OUT/trace.err || true
F=$(find $OUT/trace -name "*kernel_trace.csv" | head -1)
python3 scripts/timeline_summary.py $F > $OUT/summary.txt 2>&1
rm -rf $OUT/trace
cat $OUT/fit.log $OUT/summary.txt
