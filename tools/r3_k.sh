#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r03_k; mkdir -p $OUT; cd $R
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -x -q -k "convdet or scores or deferred or riders or planted" > $OUT/pytest.txt 2>&1; echo "pytest rc=$?" >> $OUT/pytest.txt
run() { name=$1; shift; env "$@" python bench.py --no-cpu-baseline > $OUT/bench_$name.json 2>> $OUT/bench.err; python -c "
import json,sys; d=json.loads(open('$OUT/bench_$name.json').read().strip().splitlines()[-1]); r=d['roofline']; print('%-22s'%'$name', d['value'], d['ms_per_step'], 'fwd_only', d.get('forward_only_ms_per_step'), d.get('post_processing','')[:30], r['kernel'][:20], r['avg_launch_ms'], d['clocks']['before']['gfxclk_mhz'])"; }
run ride A=1
run ride_noprobe SQDET_BENCH_NO_PROBE=1
run ride2 A=1
tail -3 $OUT/pytest.txt
python tools/kbench.py --help 2>&1 | head -5
