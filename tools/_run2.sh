cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r03_t; mkdir -p $OUT
for c in sqdet_train_fp32 sqdet_train_fp16; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/ks_$c -o ks --output-format csv -- python $R/bench.py --config $c --no-cpu-baseline --no-graph > $OUT/kstats_$c.log 2>&1
  python $R/profiles/summarize.py $(find $OUT/ks_$c -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_$c.txt "rocprofv3 --kernel-trace --stats -- python bench.py --config $c --no-cpu-baseline --no-graph" >> $OUT/kstats_$c.log 2>&1
  rm -rf $OUT/ks_$c
done
