#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r03_m; mkdir -p $OUT; cd $R
timeout 900 python -m pytest tests/test_gpu_torch_ops.py tests/test_gpu_ops.py tests/test_gpu_train.py -m gpu -x -q > $OUT/pytest.txt 2>&1; echo "pytest rc=$?" >> $OUT/pytest.txt
tail -6 $OUT/pytest.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/t -o t --output-format csv -- python $R/tools/fire1x1_standalone.py --plan $OUT/plan.json > $OUT/events_table.txt 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT/f -o f --output-format csv -- python $R/tools/fire1x1_standalone.py > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/w -o w --output-format csv -- python $R/tools/fire1x1_standalone.py > /dev/null 2>&1
python $R/tools/fire1x1_standalone.py --summarize $OUT/plan.json $(find $OUT/t -name "*kernel_trace.csv" | head -1) $(find $OUT/f -name "*counter_collection.csv" | head -1) $(find $OUT/w -name "*counter_collection.csv" | head -1) > $OUT/fire_1x1_standalone.txt 2> $OUT/summarize.err
rm -rf $OUT/t $OUT/f $OUT/w
cat $OUT/fire_1x1_standalone.txt | cut -c1-170; tail -5 $OUT/summarize.err
