#!/bin/bash
# Round-4 additions to the committed profiles (run on the GPU box through gpurun, after scripts/profile_round.sh r04):
#   r04_sq_counters_ppc.txt          SQ counters of the posterior-predictive kernels at the bench's workload (scripts/gpu_ppc_bench.py)
#   r04_kernel_stats_shards_rccl.csv rocprofv3 kernel statistics of cfg4 as one shard through the RCCL path (one rank): the row of
#                                    RCCL's all-reduce kernel is what a collective per leapfrog costs on the compute stream
set -e
export TMPDIR=/tmp
OUT=gpurun_out/prof_r04x; mkdir -p $OUT
SCRIPT=scripts/gpu_ppc_bench.py KERNEL=ppc_wave bash scripts/gpu_sq_pmc.sh ppc_r04 1 2 > $OUT/sq_ppc.log 2>&1 || true
cp gpurun_out/sqpmc_ppc_r04/summary.txt $OUT/sq_counters_ppc_wave.txt || true
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --mode shards --exchange rccl --genes 50000 --samples 500 --chains-per-gpu 4 --steps 1 --warmup 0 --no-cpu-baseline --as-named-steps 0 --nuts-warmup 30 --draws-per-chain 20 > $OUT/shards_rccl.json 2> $OUT/trace.err || true
find $OUT/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats_shards_rccl.csv || true
rm -rf $OUT/trace
head -12 $OUT/kernel_stats_shards_rccl.csv; cat $OUT/sq_counters_ppc_wave.txt
