#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r03_b; mkdir -p $OUT; cd $R
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_gpu_train.py -m gpu -x -q -k "convdet or scores or planted or detect_filter or pipelined or decision or loss or full_config or demo" > $OUT/pytest.txt 2>&1; echo "pytest rc=$?" >> $OUT/pytest.txt
python __graft_entry__.py smoke > $OUT/smoke.txt 2>&1; echo "smoke rc=$?" >> $OUT/smoke.txt
python bench.py --no-cpu-baseline > $OUT/bench_a.json 2>> $OUT/bench.err
SQDET_SCORE_EPILOGUE=0 python bench.py --no-cpu-baseline > $OUT/bench_noepi.json 2>> $OUT/bench.err
python bench.py --no-cpu-baseline > $OUT/bench_b.json 2>> $OUT/bench.err
SQDET_SCORE_EPILOGUE=0 python bench.py --no-cpu-baseline > $OUT/bench_noepi2.json 2>> $OUT/bench.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_20.json 2>> $OUT/bench.err
tail -30 $OUT/pytest.txt; cat $OUT/smoke.txt | tail -5
for f in bench_a bench_noepi bench_b bench_noepi2 bench_20; do python -c "
import json,sys; d=json.loads(open('$OUT/$f.json').read().strip().splitlines()[-1]); r=d['roofline']; print('$f', d['value'], d['ms_per_step'], 'fwd_only', d.get('forward_only_ms_per_step'), 'epi', d.get('score_epilogue'), r['kernel'], r['avg_launch_ms'], r['frac'], d['clocks']['before']['gfxclk_mhz'])"; done
tail -c 1500 $OUT/bench.err
