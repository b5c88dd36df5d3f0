#!/bin/bash
# Same-box A/B of the headline step over ENVIRONMENT variants: alternates bench.py runs, N rounds.
#   gpurun -- 'bash tools/ab_env.sh 2 "" "SQDET_SERVE_LANES=1" "SQDET_SERVE_LANES=3"'
N=${1:-2}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
for i in $(seq 1 $N); do
  for v in "$@"; do
    env $v python $R/bench.py --no-cpu-baseline --steps ${STEPS:-100} --warmup ${WARMUP:-10} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('%-28s step %.4f ms  %.0f img/s  box_mfma %s TF/s' % ('${v:-(default)}', d['ms_per_step'], d['value'], (d.get('box') or {}).get('box_mfma_tflops')))"
  done
done
