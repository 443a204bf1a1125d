#!/bin/bash
# DEV-ONLY: VGPRs / scratch / occupancy of the kernels in libhydro, one line per (kernel family, figures).
#   scripts/resources.sh [extra hipcc flags]
cd "$(dirname "$0")/.." || exit 1
hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -shared -fno-slp-vectorize -ffp-contract=on "$@" \
      -Rpass-analysis=kernel-resource-usage -o /dev/null silver2_isaacsim_amd/csrc/hydro_kernels.hip 2>&1 |
  grep -E "Function Name:| VGPRs:|ScratchSize|Occupancy" |
  sed -E 's/.*remark: +//; s/ \[-Rpass.*//; s/Function Name: _ZN12_GLOBAL__N_1[0-9]+([a-z_0-9]+kernel).*/\1/' |
  paste - - - - | sort | uniq -c
