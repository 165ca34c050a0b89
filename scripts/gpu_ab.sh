#!/bin/bash
# development aid: the log-likelihood kernel of two builds of the library, timed alternately on one box
# usage: scripts/gpu_ab.sh libA.so libB.so [rounds]   (environment of scripts/gpu_kbench.py applies: CHAINS, LANES, G, S)
A=$1; B=$2; N=${3:-3}
for i in $(seq $N); do
  for lib in $A $B; do
    printf "%s: " $lib; PPCX_LIB=$lib REPS=${REPS:-60} ROUNDS=${ROUNDS:-5} python3 scripts/gpu_kbench.py 2>&1 | tail -1 | sed -E 's/.*loglik us.launch (min [0-9.]+ median [0-9.]+).*/\1/'
  done
done
