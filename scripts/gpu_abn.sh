#!/bin/bash
# development aid: the log-likelihood kernel of several builds of the library, timed alternately on one box
# usage: ROUNDS_AB=3 scripts/gpu_abn.sh lib1.so lib2.so ...   (environment of scripts/gpu_kbench.py applies: CHAINS, LANES, G, S, WARM)
for i in $(seq ${ROUNDS_AB:-3}); do
  for lib in "$@"; do
    printf "%s: " $lib; PPCX_LIB=$lib REPS=${REPS:-60} ROUNDS=${ROUNDS:-3} python3 scripts/gpu_kbench.py 2>&1 | tail -1 | sed -E 's/.*loglik us.launch (min [0-9.]+ median [0-9.]+).*/\1/'
  done
done
