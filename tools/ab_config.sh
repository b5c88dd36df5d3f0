#!/bin/bash
# Same-box A/B of any bench config: bash tools/ab_config.sh <config> <rounds> <opt A> <opt B> ...
C=$1; N=$2; shift; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
for i in $(seq 1 $N); do
  for o in "$@"; do
    OPTS=""; for kv in ${o//,/ }; do OPTS="$OPTS --opt $kv"; done
    python $R/bench.py --config $C --no-cpu-baseline $OPTS 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('%-18s %-14s %9.1f img/s  step %.4f ms  box_mfma %s' % ('$C', '$o', d['value'], d['ms_per_step'], (d.get('box') or {}).get('box_mfma_tflops')))"
  done
done
