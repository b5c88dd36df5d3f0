#!/bin/bash
# Builds squeezedet_amd/libsqdet_hip_alt.so = the current objects with ONE source recompiled under extra defines (same-box A/B of a
# compile-time variant through SQDET_LIB):   bash tools/build_alt.sh convdet.hip "-DSQDET_CD_ASMWAIT" [-ffp-contract=off ...]
set -e
R=$(cd $(dirname $0)/.. && pwd)
SRC=$1; DEFS=$2; shift 2
OBJ=$R/squeezedet_amd/csrc/build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -Wno-inline-asm -I$R/include -I$R/squeezedet_amd/csrc $DEFS "$@" -x hip -c $R/squeezedet_amd/csrc/$SRC -o /tmp/alt_${SRC%.*}.o
OBJS=$(ls $OBJ/*.o | grep -v "/${SRC%.*}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/squeezedet_amd/libsqdet_hip_alt.so $OBJS /tmp/alt_${SRC%.*}.o
ls -la $R/squeezedet_amd/libsqdet_hip_alt.so
