#!/bin/bash
# development aid: kernel-level timing (scripts/gpu_kbench.py) of the product library and of variant builds / environments
# usage: scripts/gpu_variants.sh "<name>=<ENV assignments>" ...   (PPCX_LIB=variants/x.so selects a variant build)
for spec in "$@"; do
  name=${spec%%=*}; envs=${spec#*=}
  echo "== $name ($envs)"
  env $envs python3 scripts/gpu_kbench.py 2>&1 | tail -2
done
