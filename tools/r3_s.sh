#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r03_s; mkdir -p $OUT; cd $R
timeout 600 python -m pytest tests/test_gpu_model.py -m gpu -x -q -k "sample_png" > $OUT/pytest.txt 2>&1; echo "pytest rc=$?" >> $OUT/pytest.txt
tail -15 $OUT/pytest.txt
