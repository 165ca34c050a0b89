#!/bin/bash
# rocprofv3 runs whose summaries are committed under profiles/ (run on the GPU box through gpurun).
#   1. kernel trace + stats of the default bench command (one fit)
#   2./3. PMC passes (FETCH_SIZE, WRITE_SIZE separately: TCC slots) on a shortened fit to bound the CSV size
set -e
R=${1:-r02}
OUT=gpurun_out/prof_$R
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --as-named-steps 0 --stream-group-steps 0 > $OUT/bench_under_rocprof.json 2> $OUT/trace.err || true
find $OUT/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv || true
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --nuts-warmup 10 --draws-per-chain 6 --as-named-steps 0 --stream-group-steps 0 > /dev/null 2> $OUT/pmc_fetch.err || true
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --nuts-warmup 10 --draws-per-chain 6 --as-named-steps 0 --stream-group-steps 0 > /dev/null 2> $OUT/pmc_write.err || true
PPCX_PROFILE_CHAINS=$(python3 -c "import json;print(json.load(open('$OUT/bench_under_rocprof.json'))['config']['chains_total'])") python3 scripts/summarise_pmc.py $OUT > $OUT/pmc_summary.txt 2>&1 || true
ls -la $OUT
# keep only the small summaries in gpurun_out (the raw traces exceed the merge limit)
rm -rf $OUT/trace $OUT/pmc_fetch $OUT/pmc_write
