#!/bin/bash
# Gene-shard records of a round (run on the GPU box through gpurun):
#   rNN_bench_cfg4_shards_1gpu.json         BASELINE cfg4 (50 000 x 500, 4 chains) as ONE shard: pipelined rounds, no exchange takes place
#   rNN_bench_cfg4_shards_2ranks_1gpu.json  the same with the genes over TWO ranks that share the one GPU (gloo for the host-side
#                                           collectives, IPC handles for the exchange buffers): the direct exchange inside the merged
#                                           launch; exchange_us_per_round = what a chain's state machine waits for its peer, measured in
#                                           the kernel -- an upper bound for ranks that own a GPU each (here the two ranks' kernels
#                                           time-share the chip)
#   rNN_bench_cfg4_shards_rccl_1gpu.json    one rank through the RCCL path (three-launch round + an all-reduce per leapfrog)
set -e
R=${1:-r04}
OUT=gpurun_out/shards_$R
mkdir -p $OUT
A="--mode shards --genes 50000 --samples 500 --chains-per-gpu 4 --steps 1 --warmup 0 --no-cpu-baseline --as-named-steps 0"
python3 bench.py $A 2> $OUT/one.err | grep '^{' > $OUT/${R}_bench_cfg4_shards_1gpu.json
PPCX_DIST_BACKEND=gloo python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 $A 2> $OUT/two.err | grep '^{' > $OUT/${R}_bench_cfg4_shards_2ranks_1gpu.json
python3 bench.py $A --exchange rccl 2> $OUT/rccl.err | grep '^{' > $OUT/${R}_bench_cfg4_shards_rccl_1gpu.json
for f in $OUT/*.json; do python3 - $f <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); c = d["config"]
print(sys.argv[1].split("/")[-1], "value", d["value"], "ms_per_step", d["ms_per_step"], "n_gpus", d["n_gpus"], "exchange", c.get("exchange"), "exchange_us_per_round", c.get("exchange_us_per_round"),
      "roofline", (d["roofline"] or {}).get("frac"), "launch ms", (d["roofline"] or {}).get("avg_launch_ms"), "round", c.get("round_structure", "")[:20])
PY
done
