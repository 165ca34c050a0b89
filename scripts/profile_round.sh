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
# the merged launch's durations in that trace: the statistics' mean contains the partly empty launches after chains have finished; the
# median is a launch with every chain active -- what bench.py's roofline sample (events attached to such dispatches) must agree with
python3 - $OUT <<'PY' || true
import csv, glob, json, sys
import numpy as np
out = sys.argv[1]
f = glob.glob(out + "/trace/**/*kernel_trace.csv", recursive=True)[0]
d = np.array([int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(f)) if "ppcx_ls_kernel" in r["Kernel_Name"]], float) * 1e-3
bench = json.load(open(out + "/bench_under_rocprof.json"))
json.dump({"kernel": "ppcx_ls_kernel", "dispatches": int(d.size), "mean_us": round(float(d.mean()), 2), "median_us": round(float(np.median(d)), 2),
           "p25_us": round(float(np.percentile(d, 25)), 2), "p75_us": round(float(np.percentile(d, 75)), 2),
           "bench_roofline_avg_launch_us_same_run": round(1e3 * bench["roofline"]["avg_launch_ms"], 2)}, open(out + "/ls_kernel_durations.json", "w"), indent=1)
PY
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
