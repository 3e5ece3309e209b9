#!/bin/bash
# Prints VGPR/SGPR/LDS/occupancy per kernel (hipcc -Rpass-analysis=kernel-resource-usage).
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -ffp-contract=off -x hip -c minnow_amd/csrc/mnw_kernels.hip -o /tmp/mnw_k.o \
  -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "Function Name|VGPRs:|AGPRs|SGPRs:|Occupancy|LDS Size|ScratchSize" \
  | sed -E 's/.*remark: [^ ]+ +//; s/ \[-Rpass.*//; s/.*kernel-resource-usage\]//' | sed -E 's/^.*: +(Function Name)/\n\1/' | awk '{printf "%s  ", $0} /LDS Size/ {print ""}'
