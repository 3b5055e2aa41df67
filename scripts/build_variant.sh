#!/bin/bash
# Development aid: libegogen_hip.so with extra -D flags on body_model.hip only -> ab_libs/lib_<name>.so (the other objects as built)
#   bash scripts/build_variant.sh <name> "-DEGX_LBS_HOIST_MAP=1 ..." [report]
set -eu
R=$(cd "$(dirname "$0")/.." && pwd)
N=$1; FL=${2:-}; REP=${3:-}
mkdir -p $R/ab_libs $R/build
make -s -C $R/egogen_amd/csrc > /dev/null
EXTRA=""
[ -n "$REP" ] && EXTRA="-Rpass-analysis=kernel-resource-usage"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function $FL $EXTRA -c $R/egogen_amd/csrc/body_model.hip -o $R/build/body_model_$N.o 2> $R/build/body_model_$N.log || { tail -20 $R/build/body_model_$N.log; exit 1; }
OBJS=$(ls $R/egogen_amd/csrc/*.o | grep -v body_model.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS $R/build/body_model_$N.o -o $R/ab_libs/lib_$N.so
ls -la $R/ab_libs/lib_$N.so
