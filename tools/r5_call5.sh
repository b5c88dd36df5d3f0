#!/bin/bash
# GPU call 5 of round 5: loop-structure microbenchmark of the streaming 1x1; ConvDet with the compiler-visible tile wait: parity + same-box A/B
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=gpurun_out/r05e
mkdir -p $O
hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form=1 -o /tmp/c1_structure tools/microbench/c1_structure.hip && timeout 200 /tmp/c1_structure > $O/c1_structure.txt 2>&1
cat $O/c1_structure.txt
timeout 600 python -m pytest tests/test_gpu_model.py tests/test_gpu_ops.py -m gpu -x -q -k "chain or convdet or headline or pipelined or plan or smoke or conv" > $O/pytest_subset.log 2>&1
tail -3 $O/pytest_subset.log
bash tools/ab_lib.sh 3 squeezedet_amd/libsqdet_hip_alt.so > $O/ab_cdwait_2lanes.txt 2>&1
cat $O/ab_cdwait_2lanes.txt
SQDET_SERVE_LANES=1 bash tools/ab_lib.sh 3 squeezedet_amd/libsqdet_hip_alt.so > $O/ab_cdwait_1lane.txt 2>&1
cat $O/ab_cdwait_1lane.txt
