#!/bin/bash
# Same-box A/B of a training config: alternates bench.py runs over "ENV=.. --opt .." variants, N rounds.
#   gpurun -- 'bash tools/ab_train.sh res50_train_fp16 2 "SQDET_FUSE_RELU_BWD=0 dbg=52" "SQDET_FUSE_RELU_BWD=1 dbg=0"'
# (each variant: environment assignments, then knob=value options)
CFG=$1; N=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}
for i in $(seq 1 $N); do
  for v in "$@"; do
    ENVS=""; OPTS=""
    for tok in $v; do
      case $tok in
        SQDET_*) ENVS="$ENVS $tok";;
        *) OPTS="$OPTS --opt $tok";;
      esac
    done
    env $ENVS python $R/bench.py --config $CFG --no-cpu-baseline --steps ${STEPS:-60} --warmup ${WARMUP:-10} $OPTS 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('%-40s %s  %.1f %s  step %.4f ms' % ('$v', d['config']['name'], d['value'], d['unit'], d['ms_per_step']))"
  done
done
