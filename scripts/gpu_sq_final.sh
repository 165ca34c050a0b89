#!/bin/bash
# development aid: SQ counter passes of the log-likelihood workgroups (kernel-level bench) for profiles/rNN_sq_counters_loglik.txt
export KERNEL=loglik REPS=20 ROUNDS=2 LANES=${LANES:-0}    # lanes per gene: the library's choice for 8 chains
scripts/gpu_sq_pmc.sh final 1 2 3 > gpurun_out/sq_final.txt 2>&1
ROUNDS=4 REPS=100 LANES=$LANES python3 scripts/gpu_kbench.py 2>&1 | tail -1 > gpurun_out/kbench_final.txt
cat gpurun_out/kbench_final.txt; tail -30 gpurun_out/sq_final.txt
