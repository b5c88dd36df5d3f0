mkdir -p gpurun_out/r03_w4
timeout 1200 python -m pytest tests/test_gpu_train.py tests/test_gpu_resnet.py tests/test_gpu_dist.py -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r03_w4/tests.txt
for cfg in sqdet_train_fp16 sqdet_train_fp32 res50_train_fp16; do
  timeout 300 python bench.py --config $cfg --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r03_w4/${cfg}.json
done
cat gpurun_out/r03_w4/tests.txt
for f in gpurun_out/r03_w4/*.json; do echo $f; python -c "import json,sys; d=json.load(open('$f')); print(d['value'], d['ms_per_step'])"; done
