#!/bin/bash
# development aid: time the loglik kernel for every library in variants/ (built with different -D switches)
out=gpurun_out/variants.log; : > $out
for lib in variants/*.so; do
  for st in ${STAGES:-0 32}; do
    PPCX_LIB=$PWD/$lib PPCX_STAGE_CELLS=$st timeout -k 10 120 python scripts/gpu_kbench.py | sed "s|^|$lib stage=$st |" >> $out || exit 1
  done
done
cat $out
