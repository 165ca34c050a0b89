#!/bin/bash
# The per-round records committed under profiles/ besides the rocprofv3 summaries (run on the GPU box through gpurun):
#   rNN_bench_default.json            the default bench line (what the driver runs)
#   (rNN_bench_cfg4_shards_*.json: scripts/gpu_shard_records.sh)
#   rNN_cfg5_two_pass.json            BASELINE config 5 end to end through identify_outliers()
#   rNN_advi_cfg3.json                the reference's default inference mode (ADVI) at cfg3 size
# Only lines that start with "{" are kept: RCCL prints its version banner on stdout.
set -e
R=${1:-r03}
OUT=gpurun_out/records_$R
mkdir -p $OUT
python3 bench.py --steps 20 --warmup 5 2> $OUT/bench_default.err | grep '^{' > $OUT/${R}_bench_default.json    # as the driver runs it
# (the gene-shard records: scripts/gpu_shard_records.sh)
python3 scripts/gpu_cfg5.py 2> $OUT/cfg5.err | grep '^{' > $OUT/${R}_cfg5_two_pass.json
python3 scripts/gpu_advi_time.py 2> $OUT/advi.err | grep '^{' > $OUT/${R}_advi_cfg3.json
ls -la $OUT
