#!/bin/bash
# VGPRs / spills / scratch / occupancy of the kernels of one csrc/*.hip file whose mangled name matches a pattern (hipcc remarks).
# usage: scripts/kernel_resources.sh body_model.hip [pattern] [extra hipcc flags...]
cd "$(dirname "$0")/../egogen_amd/csrc"
f=$1; pat=${2:-.}; shift; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Rpass-analysis=kernel-resource-usage "$@" -c $f -o /tmp/_kr.o 2>&1 |
  grep -E "remark:" | sed -E 's/.*remark: +//; s/ \[-Rpass.*//' |
  awk '/Function Name/ {if (line) print line; line=$3; next} /VGPRs:|Spill|ScratchSize|Occupancy/ {line=line "  " $0} END {print line}' |
  grep -E "$pat" | sed -E 's/ +/ /g'
