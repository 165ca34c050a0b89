#!/bin/bash
# development aid: rocprofv3 kernel statistics of a short cfg3 fit (8 chains, 30 + 30 iterations) in the mode given by the environment
# usage: PPCX_PIPELINE=1 scripts/gpu_prof_short.sh <tag>
export TMPDIR=/tmp
TAG=${1:-run}
OUT=gpurun_out/profshort_$TAG; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 scripts/gpu_fit_short.py > $OUT/fit.log 2> $OUT/trace.err || true
find $OUT/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv || true
rm -rf $OUT/trace
cat $OUT/fit.log; cut -c1-160 $OUT/kernel_stats.csv | head -6
