#!/bin/bash
# The stand-alone fire 1x1 table (profiles/rNN_fire_1x1_standalone.txt): rocprofv3 kernel trace + FETCH_SIZE / WRITE_SIZE passes of
# tools/fire1x1_standalone.py, summarised.    gpurun -- 'bash tools/collect_fire1x1.sh r04_x'
set -u
TAG=${1:-r04_f1}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $OUT/kt -o t --output-format csv -- python $R/tools/fire1x1_standalone.py --plan $OUT/plan.json > $OUT/events.txt 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT/f -o f --output-format csv -- python $R/tools/fire1x1_standalone.py > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/w -o w --output-format csv -- python $R/tools/fire1x1_standalone.py > /dev/null 2>&1
cd $R
python tools/fire1x1_standalone.py --summarize $OUT/plan.json $(find $OUT/kt -name "*kernel_trace.csv" | head -1) $(find $OUT/f -name "*counter_collection.csv" | head -1) $(find $OUT/w -name "*counter_collection.csv" | head -1) > $OUT/fire_1x1_standalone.txt 2> $OUT/summarize.err
rm -rf $OUT/kt $OUT/f $OUT/w
tail -25 $OUT/fire_1x1_standalone.txt; tail -3 $OUT/summarize.err
