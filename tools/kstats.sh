#!/bin/bash
# rocprofv3 kernel stats of bench.py configs into gpurun_out/$1/kernel_stats_<config>.txt      gpurun -- 'bash tools/kstats.sh r06_x "res50_train_fp16 sqdetplus_infer"'
set -u
TAG=${1:-r06_ks}; CONFIGS=${2:-"res50_train_fp16"}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
FP=$(cd $R && python -c "import bench; print(bench.build_fingerprint())")
for c in $CONFIGS; do
  rocprofv3 --kernel-trace --stats -d $OUT/ks_$c -o ks --output-format csv -- python $R/bench.py --config $c --no-cpu-baseline --no-graph > $OUT/kstats_$c.log 2>&1
  python $R/profiles/summarize.py $(find $OUT/ks_$c -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_$c.txt "rocprofv3 --kernel-trace --stats -- python bench.py --config $c --no-cpu-baseline --no-graph" $FP >> $OUT/kstats_$c.log 2>&1
  rm -rf $OUT/ks_$c
done
