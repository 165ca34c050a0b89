#!/bin/bash
# development aid: register / scratch / LDS use of every kernel in ppcx_kernels.hip (hipcc resource-usage remarks)
# usage: scripts/kres.sh [extra hipcc flags]
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -c -Rpass-analysis=kernel-resource-usage "$@" ppcseq_amd/csrc/ppcx_kernels.hip -o /tmp/kres.o 2>&1 \
 | grep -E "Function Name|TotalSGPRs|VGPRs:|Spill|ScratchSize|Occupancy|LDS Size" | sed -E 's/.*remark: [^ ]+ +//; s/ \[-Rpass.*//' | paste - - - - - - - - \
 | awk -F'\t' '{n=$1; sub(/Function Name: /,"",n); printf "%-62s %s | %s | %s | %s | %s | %s\n", substr(n,1,62), $3,$4,$5,$6,$7,$8}'
