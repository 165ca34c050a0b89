#!/bin/bash
# rocprofv3 runs whose summaries are committed under profiles/ (run on the GPU box through gpurun).
#   1. kernel trace + stats of the bench command on ONE in-order stream (--stream-groups 1): every launch of the merged
#      log-likelihood / state-machine kernel then covers all 8 chains, which is what the bench line's roofline sample times
#      (bench.py takes that sample from a single-stream fit also in its default run); the posterior-predictive kernel's
#      launches of the `ppc` object are in the same statistics
#   2. the same with the library's default (chain groups on their own streams): what `value` is measured on
#   3./4. PMC passes (FETCH_SIZE, WRITE_SIZE separately: TCC slots) on a shortened fit to bound the CSV size
set -e
R=${1:-r03}
OUT=gpurun_out/prof_$R
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --as-named-steps 0 --stream-groups 1 --single-stream-steps 1 > $OUT/bench_under_rocprof.json 2> $OUT/trace.err || true
find $OUT/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv || true
rm -rf $OUT/trace
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace2 -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --as-named-steps 0 --single-stream-steps 0 --no-ppc > $OUT/bench_default_groups_under_rocprof.json 2> $OUT/trace2.err || true
find $OUT/trace2 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats_default_groups.csv || true
rm -rf $OUT/trace2
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --nuts-warmup 10 --draws-per-chain 6 --as-named-steps 0 --stream-groups 1 --single-stream-steps 0 > /dev/null 2> $OUT/pmc_fetch.err || true
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --nuts-warmup 10 --draws-per-chain 6 --as-named-steps 0 --stream-groups 1 --single-stream-steps 0 > /dev/null 2> $OUT/pmc_write.err || true
PPCX_PROFILE_CHAINS=$(python3 -c "import json;print(json.load(open('$OUT/bench_under_rocprof.json'))['config']['chains_total'])") python3 scripts/summarise_pmc.py $OUT > $OUT/pmc_summary.txt 2>&1 || true
ls -la $OUT
# keep only the small summaries in gpurun_out (the raw traces exceed the merge limit)
rm -rf $OUT/pmc_fetch $OUT/pmc_write
