#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r03_e; mkdir -p $OUT; cd $R
run() { name=$1; shift; env "$@" python bench.py --no-cpu-baseline > $OUT/bench_$name.json 2>> $OUT/bench.err; python -c "
import json,sys; d=json.loads(open('$OUT/bench_$name.json').read().strip().splitlines()[-1]); r=d['roofline']; print('%-22s'%'$name', d['value'], d['ms_per_step'], 'fwd_only', d.get('forward_only_ms_per_step'), 'epi', d.get('score_epilogue'), r['kernel'][:20], r['avg_launch_ms'], d['clocks']['before']['gfxclk_mhz'])"; }
run dyn_epi A=1
run static_epi SQDET_OPTIONS=dbg=200
run dyn_noepi SQDET_SCORE_EPILOGUE=0
run static_noepi SQDET_OPTIONS=dbg=200 SQDET_SCORE_EPILOGUE=0
run dyn_epi2 A=1
run static_epi2 SQDET_OPTIONS=dbg=200
