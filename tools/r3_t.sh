#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r03_t; mkdir -p $OUT; cd $R
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_ops.py tests/test_gpu_torch_ops.py -m gpu -x -q -k "train or fire or step or pack or backward" > $OUT/pytest.txt 2>&1; echo "pytest rc=$?" >> $OUT/pytest.txt
tail -6 $OUT/pytest.txt
for c in sqdet_train_fp16 sqdet_train_fp32; do
  python bench.py --config $c --no-cpu-baseline > $OUT/bench_$c.json 2> $OUT/bench_$c.err
  python -c "
import json; d=json.loads(open('$OUT/bench_$c.json').read().strip().splitlines()[-1]); print('$c', d.get('value'), d.get('ms_per_step'), d['clocks']['before'].get('gfxclk_mhz'), d.get('losses'))"
done
