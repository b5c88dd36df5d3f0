#!/bin/bash
# first GPU call of round 5: parity tests, box calibration with counters, FAST profile collection, chain timeline
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out/r05a
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r05a/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r05a/pytest_gpu.log
tail -5 gpurun_out/r05a/pytest_gpu.log
timeout 300 bash tools/calib_pmc.sh r05a_cal > gpurun_out/r05a/calib.log 2>&1
FAST=1 timeout 600 bash tools/collect_profiles.sh r05a > gpurun_out/r05a/collect.log 2>&1
cat gpurun_out/r05a/bench_sqdet_infer.json
# chain timeline: chain.hip rebuilt with the stamps (only that object), then restored
touch squeezedet_amd/csrc/chain.hip
SQDET_EXTRA_DEFINES="-DSQDET_CHAIN_TIMELINE" python -m squeezedet_amd.build > gpurun_out/r05a/tl_build.log 2>&1
timeout 300 python tools/chain_timeline.py > gpurun_out/r05a/chain_timeline.txt 2>&1
cat gpurun_out/r05a/chain_timeline.txt
