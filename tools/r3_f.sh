#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r03_f; mkdir -p $OUT; cd $R
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -x -q -k "detect_filter or filter or deferred or pipelined or planted or convdet or plan_scores" > $OUT/pytest.txt 2>&1; echo "pytest rc=$?" >> $OUT/pytest.txt
run() { name=$1; shift; env "$@" python bench.py --no-cpu-baseline > $OUT/bench_$name.json 2>> $OUT/bench.err; python -c "
import json,sys; d=json.loads(open('$OUT/bench_$name.json').read().strip().splitlines()[-1]); r=d['roofline']; print('%-22s'%'$name', d['value'], d['ms_per_step'], 'fwd_only', d.get('forward_only_ms_per_step'), 'epi', d.get('score_epilogue'), 'defer', d.get('deferred_post'), r['kernel'][:20], r['avg_launch_ms'], d['clocks']['before']['gfxclk_mhz'])"; }
run defer16 A=1
run nodefer SQDET_POST_DEFER=0
run defer32 SQDET_POST_WGS=32
run defer8 SQDET_POST_WGS=8
run defer16_prio0 SQDET_POST_PRIORITY=0
run defer16_b A=1
run nodefer_b SQDET_POST_DEFER=0
run defer16_20steps A=1
tail -4 $OUT/pytest.txt
