#!/bin/bash
# GPU call 3 of round 5: store-shape / read:write microbenchmarks, parity of the L2 warm-up build, same-box A/B of it
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=gpurun_out/r05c
mkdir -p $O
hipcc --offload-arch=gfx950 -O3 -o /tmp/store_shape tools/microbench/store_shape.hip && timeout 120 /tmp/store_shape > $O/store_shape.txt 2>&1
cat $O/store_shape.txt
hipcc --offload-arch=gfx950 -O3 -o /tmp/rw_ceiling tools/microbench/rw_ceiling.hip && timeout 120 /tmp/rw_ceiling > $O/rw_ceiling.txt 2>&1
timeout 600 python -m pytest tests/test_gpu_model.py tests/test_gpu_ops.py -m gpu -x -q -k "chain or convdet or headline or pipelined or plan or smoke" > $O/pytest_subset.log 2>&1
tail -4 $O/pytest_subset.log
STEPS=100 bash tools/ab_bench.sh 3 dbg=90 dbg=0 > $O/ab_warm_2lanes.txt 2>&1
cat $O/ab_warm_2lanes.txt
SQDET_SERVE_LANES=1 STEPS=100 bash tools/ab_bench.sh 3 dbg=90 dbg=0 > $O/ab_warm_1lane.txt 2>&1
cat $O/ab_warm_1lane.txt
