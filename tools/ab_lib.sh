#!/bin/bash
# Same-box A/B of two builds of the library (a compile-time variant): alternates bench.py under SQDET_LIB=<alt> and the default build.
#   gpurun -- 'bash tools/ab_lib.sh 3 squeezedet_amd/libsqdet_hip_alt.so'      (SQDET_SERVE_LANES, STEPS as for ab_bench.sh)
N=${1:-3}; ALT=$2
R=${GRAFT_REPO_ROOT:-$(pwd)}
for i in $(seq 1 $N); do
  for which in alt default; do
    if [ $which = alt ]; then export SQDET_LIB=$R/$ALT; else unset SQDET_LIB; fi
    python $R/bench.py --no-cpu-baseline --steps ${STEPS:-100} --warmup ${WARMUP:-10} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
sl=(d.get('roofline') or {}).get('single_lane') or {}
print('%-8s lanes %s step %.4f ms  fwd-only %.4f  dominant launch %.4f ms (1 lane: %s)  box_mfma %s TF/s  clk %s' % ('$which', (d.get('pipeline') or {}).get('forwards_in_flight'), d['ms_per_step'], d['forward_only_ms_per_step'], d['roofline']['avg_launch_ms'], sl.get('avg_launch_ms'), (d.get('box') or {}).get('box_mfma_tflops'), d['clocks']['before'].get('gfxclk_mhz')))"
  done
done
