#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r03_j; mkdir -p $OUT; cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.txt 2>&1; echo "pytest rc=$?" >> $OUT/pytest.txt
python __graft_entry__.py smoke > $OUT/smoke.txt 2>&1; echo "smoke rc=$?" >> $OUT/smoke.txt
run() { name=$1; shift; env "$@" python bench.py --no-cpu-baseline > $OUT/bench_$name.json 2>> $OUT/bench.err; python -c "
import json,sys; d=json.loads(open('$OUT/bench_$name.json').read().strip().splitlines()[-1]); r=d['roofline']; print('%-22s'%'$name', d['value'], d['ms_per_step'], 'fwd_only', d.get('forward_only_ms_per_step'), d.get('post_processing','')[:30], r['kernel'][:20], r['avg_launch_ms'], d['clocks']['before']['gfxclk_mhz'])"; }
run ride A=1
run ride_noprobe SQDET_BENCH_NO_PROBE=1
run nodefer SQDET_POST_DEFER=0
run ride2 A=1
run ride_noprobe2 SQDET_BENCH_NO_PROBE=1
tail -4 $OUT/pytest.txt; tail -3 $OUT/smoke.txt
