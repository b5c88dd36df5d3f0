#!/bin/bash
# Same-box A/B of the headline step: alternates `bench.py --opt $A` and `bench.py --opt $B` N times and prints ms_per_step of each run.
#   gpurun -- 'bash tools/ab_bench.sh "dbg=70" "dbg=0" 3'
A=${1:-dbg=0}; B=${2:-dbg=0}; N=${3:-3}
R=${GRAFT_REPO_ROOT:-$(pwd)}
for i in $(seq 1 $N); do
  for o in "$A" "$B"; do
    python $R/bench.py --no-cpu-baseline --steps 100 --warmup 10 --opt $o 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('%-10s step %.4f ms  fwd-only %.4f  box_mfma %s TF/s  copy %s GB/s  clk %s' % ('$o', d['ms_per_step'], d['forward_only_ms_per_step'], (d.get('box') or {}).get('box_mfma_tflops'), (d.get('box') or {}).get('box_copy_gbs'), d['clocks']['before'].get('gfxclk_mhz')))"
  done
done
