#!/bin/bash
# round 3, first GPU call: state probe, the gpu test-suite, the three inference bench lines
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r03_a; mkdir -p $OUT; cd $R
bash tools/probe_gpu_state.sh > $OUT/probe.txt 2>&1
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.txt 2>&1; echo "pytest rc=$?" >> $OUT/pytest.txt
python bench.py > $OUT/bench_sqdet_infer.json 2> $OUT/bench.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_sqdet_infer_20.json 2>> $OUT/bench.err
python bench.py --config sqdet_infer_384 --no-cpu-baseline > $OUT/bench_sqdet_infer_384.json 2>> $OUT/bench.err
python bench.py --config sqdet_sample_b1 > $OUT/bench_sqdet_sample_b1.json 2>> $OUT/bench.err
python bench.py --gpus 2 > $OUT/bench_gpus2_on_1gpu_box.json 2>> $OUT/bench.err; echo "rc=$?" >> $OUT/bench_gpus2_on_1gpu_box.json
tail -5 $OUT/pytest.txt; tail -c 600 $OUT/bench_sqdet_infer.json; echo; tail -c 3000 $OUT/bench.err
