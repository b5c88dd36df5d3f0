#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r03_o; mkdir -p $OUT; cd $R
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -x -q -k "stem or plan or planted or layer_by_layer or full_config" > $OUT/pytest.txt 2>&1; echo "pytest rc=$?" >> $OUT/pytest.txt
tail -5 $OUT/pytest.txt
run() { name=$1; shift; env "$@" python bench.py --no-cpu-baseline > $OUT/bench_$name.json 2>> $OUT/bench.err; python -c "
import json,sys; d=json.loads(open('$OUT/bench_$name.json').read().strip().splitlines()[-1]); r=d['roofline']; print('%-22s'%'$name', d['value'], d['ms_per_step'], 'fwd_only', d.get('forward_only_ms_per_step'), r['kernel'][:20], r['avg_launch_ms'], r['frac'], d['clocks']['before']['gfxclk_mhz'])"; }
run phase A=1
run pers SQDET_OPTIONS=stem_algo=3
run phase2 A=1
run pers2 SQDET_OPTIONS=stem_algo=3
