#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r03_c; mkdir -p $OUT; cd $R
python __graft_entry__.py smoke > $OUT/smoke.txt 2>&1; echo "smoke rc=$?" >> $OUT/smoke.txt
run() { name=$1; shift; env "$@" python bench.py --no-cpu-baseline > $OUT/bench_$name.json 2>> $OUT/bench.err; python -c "
import json,sys; d=json.loads(open('$OUT/bench_$name.json').read().strip().splitlines()[-1]); r=d['roofline']; print('%-22s'%'$name', d['value'], d['ms_per_step'], 'fwd_only', d.get('forward_only_ms_per_step'), 'epi', d.get('score_epilogue'), r['kernel'][:20], r['avg_launch_ms'], d['clocks']['before']['gfxclk_mhz'])"; }
run epi_side A=1
run noepi_side SQDET_SCORE_EPILOGUE=0
run epi_inline SQDET_POST_INLINE=1
run noepi_inline SQDET_POST_INLINE=1 SQDET_SCORE_EPILOGUE=0
run epi_side_nod2h SQDET_BENCH_NO_D2H=1
run epi_side_prio0 SQDET_POST_PRIORITY=0
run noepi_side_prio0 SQDET_POST_PRIORITY=0 SQDET_SCORE_EPILOGUE=0
run epi_inline_nod2h SQDET_POST_INLINE=1 SQDET_BENCH_NO_D2H=1
run epi_side2 A=1
tail -3 $OUT/smoke.txt
