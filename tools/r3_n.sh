#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
echo "== PPR=4"; python tools/exp_stem4.py 2>&1 | tail -7
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$R/include -I$R/squeezedet_amd/csrc -mllvm -amdgpu-mfma-vgpr-form=1 -fno-honor-nans -DSQDET_STEM4_PPR=2 -x hip -c squeezedet_amd/csrc/stem4.hip -o squeezedet_amd/csrc/build/stem4.o 2>/dev/null
hipcc --offload-arch=gfx950 -shared -fPIC -o squeezedet_amd/libsqdet_hip.so squeezedet_amd/csrc/build/*.o
echo "== PPR=2"; python tools/exp_stem4.py 2>&1 | tail -7
