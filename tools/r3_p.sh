#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r03_p; mkdir -p $OUT; cd $R
for c in sqdetplus_infer sqdet_train_fp32 res50_train_fp16 sqdet_train_fp16 sqdet_infer_384 sqdet_sample_b1; do
  python bench.py --config $c --no-cpu-baseline > $OUT/bench_$c.json 2> $OUT/bench_$c.err; echo "$c rc=$?"
  python -c "
import json; d=json.loads(open('$OUT/bench_$c.json').read().strip().splitlines()[-1]); r=d.get('roofline',{}); print('  ', d.get('value'), d.get('ms_per_step'), d.get('post_processing','')[:40], r.get('kernel','')[:30], r.get('frac'), d.get('error'), d.get('latency_ms_per_image_sync'))"
  tail -3 $OUT/bench_$c.err | grep -v amdgpu.ids
done
