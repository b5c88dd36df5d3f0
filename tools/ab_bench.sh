#!/bin/bash
# Same-box A/B of the headline step: alternates `bench.py --opt X` over the given option strings, N rounds, prints ms_per_step of each run.
#   gpurun -- 'bash tools/ab_bench.sh 3 dbg=70 dbg=0'      (an option string may hold several knobs: "stem_algo=6,dbg=70")
N=${1:-3}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
for i in $(seq 1 $N); do
  for o in "$@"; do
    OPTS=""; for kv in ${o//,/ }; do OPTS="$OPTS --opt $kv"; done
    python $R/bench.py --no-cpu-baseline --steps ${STEPS:-100} --warmup ${WARMUP:-10} $OPTS 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('%-22s lanes %s step %.4f ms  fwd-only %.4f  box_mfma %s TF/s  copy %s GB/s  clk %s' % ('$o', (d.get('pipeline') or {}).get('forwards_in_flight'), d['ms_per_step'], d['forward_only_ms_per_step'], (d.get('box') or {}).get('box_mfma_tflops'), (d.get('box') or {}).get('box_copy_gbs'), d['clocks']['before'].get('gfxclk_mhz')))"
  done
done
