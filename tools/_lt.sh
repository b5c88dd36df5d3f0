timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -x -q -k "chain or plan or layer or pipelined or planted or riders or post_job" 2>&1 | tail -2
python bench.py --no-cpu-baseline --layer-table gpurun_out/lt_new.json > gpurun_out/lt_bench.json 2>/dev/null
python - <<'PY'
import json
d=json.load(open('gpurun_out/lt_new.json'))
for l in d['layers'][4:]: print("%-45s %.2f us" % (l['layer'], l['ms']*1e3))
print(d['layers'][0]['ms']*1e3, d['forward_ms_sum'], d['step_ms'])
b=json.loads(open('gpurun_out/lt_bench.json').read().strip().split('\n')[-1]); print(b['value'], b['ms_per_step'], b['clocks']['before']['sclk_mhz'], b['clocks']['after']['sclk_mhz'])
PY
