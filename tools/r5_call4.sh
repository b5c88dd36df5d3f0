#!/bin/bash
# GPU call 4 of round 5: branch-free conv1x1_stream / conv1x1_deepk -- parity, stand-alone sweep + table (events), the configs they serve
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=gpurun_out/r05d
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_resnet.py tests/test_gpu_train.py -m gpu -x -q > $O/pytest_subset.log 2>&1
tail -4 $O/pytest_subset.log
timeout 300 python tools/c1_sweep.py > $O/c1_sweep.txt 2>&1
cat $O/c1_sweep.txt
timeout 300 python tools/fire1x1_standalone.py > $O/fire1x1_events.txt 2>&1
cat $O/fire1x1_events.txt
for c in sqdetplus_infer res50_train_fp16 sqdet_train_fp16 sqdet_train_fp32; do
  timeout 300 python bench.py --config $c --no-cpu-baseline > $O/bench_$c.json 2> $O/bench_$c.err
  python -c "import json;d=json.load(open('$O/bench_$c.json'));print('$c', d.get('value'), d.get('ms_per_step'), d.get('error'))"
done
