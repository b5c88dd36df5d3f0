#!/bin/bash
# Runs on the GPU box (gpurun): everything profiles/ needs for one build, into gpurun_out/$1/.
#   gpurun -- 'bash tools/collect_profiles.sh r03_x'          (~3 GPU-minutes)
#   gpurun -- 'FAST=1 bash tools/collect_profiles.sh r02_x'   (~1 GPU-minute: the headline's bench lines, layer table, kernel
#       stats and the fingerprinted PMC traffic only -- what bench.py's roofline.traffic needs after a csrc change)
set -u
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# 1. the bench lines (driver-style short run of the headline too)
CONFIGS="sqdet_infer sqdet_infer_384 sqdet_sample_b1 sqdetplus_infer sqdet_train_fp32 res50_train_fp16 sqdet_train_fp16"
[ "${FAST:-0}" = "1" ] && CONFIGS="sqdet_infer ${EXTRA:-}"      # EXTRA="res50_train_fp16 ..": those configs' bench lines + kernel stats too
for c in $CONFIGS; do
  python $R/bench.py --config $c > $OUT/bench_$c.json 2> $OUT/bench_$c.err
done
python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_sqdet_infer_20steps.json 2>> $OUT/bench_sqdet_infer.err
python $R/bench.py --no-cpu-baseline --layer-table $OUT/layer_table.json > /dev/null 2>&1
# 2. rocprofv3 kernel stats of the headline command
rocprofv3 --kernel-trace --stats -d $OUT/kstats -o ks --output-format csv -- python $R/bench.py --no-cpu-baseline > $OUT/kstats.log 2>&1
FP=$(cd $R && python -c "import bench; print(bench.build_fingerprint())")
python $R/profiles/summarize.py $(find $OUT/kstats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.txt "rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline" $FP >> $OUT/kstats.log 2>&1
# 2a. the same command with ONE forward in flight (SQDET_SERVE_LANES=1): per-launch durations without the other lane's launches
# sharing the chip -- what the per-launch roofline table is computed from
SQDET_SERVE_LANES=1 rocprofv3 --kernel-trace --stats -d $OUT/kstats1 -o ks --output-format csv -- python $R/bench.py --no-cpu-baseline > $OUT/kstats1.log 2>&1
python $R/profiles/summarize.py $(find $OUT/kstats1 -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_1lane.txt "SQDET_SERVE_LANES=1 rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline" $FP >> $OUT/kstats1.log 2>&1
SQDET_SERVE_LANES=1 python $R/bench.py --no-cpu-baseline > $OUT/bench_sqdet_infer_1lane.json 2>> $OUT/bench_sqdet_infer.err
rm -rf $OUT/kstats1
KS_CONFIGS="sqdetplus_infer sqdet_train_fp32 res50_train_fp16 sqdet_train_fp16"
[ "${FAST:-0}" = "1" ] && KS_CONFIGS="${EXTRA:-}"
# 2b. kernel stats of the other configs (which kernels carry SqueezeDet+, ResNet50 inference and the two training steps)
for c in $KS_CONFIGS; do
  rocprofv3 --kernel-trace --stats -d $OUT/ks_$c -o ks --output-format csv -- python $R/bench.py --config $c --no-cpu-baseline --no-graph > $OUT/kstats_$c.log 2>&1
  python $R/profiles/summarize.py $(find $OUT/ks_$c -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_$c.txt "rocprofv3 --kernel-trace --stats -- python bench.py --config $c --no-cpu-baseline --no-graph" $FP >> $OUT/kstats_$c.log 2>&1
  rm -rf $OUT/ks_$c
done
if [ "${FAST:-0}" != "1" ]; then
rocprofv3 --kernel-trace --stats -d $OUT/ks_res50_infer -o ks --output-format csv -- python $R/tools/netbench.py --arch resnet50 --batch 8 > $OUT/netbench_resnet50_b8.txt 2>&1
python $R/profiles/summarize.py $(find $OUT/ks_res50_infer -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_res50_infer.txt "rocprofv3 --kernel-trace --stats -- python tools/netbench.py --arch resnet50 --batch 8" >> $OUT/kstats.log 2>&1
rm -rf $OUT/ks_res50_infer
python $R/tools/nextrows_bench.py > $OUT/nextrows_bench.json 2>/dev/null
fi
# 3. HBM (fabric) traffic per launch: FETCH_SIZE and WRITE_SIZE in separate passes
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o fetch --output-format csv -- python $R/tools/pmc_forward.py > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o write --output-format csv -- python $R/tools/pmc_forward.py > $OUT/pmc_write.log 2>&1
LAYERS=$(python -c "import json; print(','.join(l['layer'] for l in json.load(open('$OUT/layer_table.json'))['layers']))")
cd $R && python profiles/pmc_traffic.py $(find $OUT/pmc_fetch -name "*counter_collection.csv" | head -1) $(find $OUT/pmc_write -name "*counter_collection.csv" | head -1) $OUT/hbm_traffic_pmc.json "$LAYERS" > $OUT/pmc_traffic.log 2>&1
# 4. SQ counters (two passes) of the forward
if [ "${FAST:-0}" != "1" ]; then
cd /tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $OUT/sq1 -o sq1 --output-format csv -- python $R/tools/pmc_forward.py > /dev/null 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $OUT/sq2 -o sq2 --output-format csv -- python $R/tools/pmc_forward.py > /dev/null 2>&1
cd $R
for f in $(find $OUT/sq1 $OUT/sq2 -name "*counter_collection.csv"); do python tools/pmc_summary.py $f sqdet; done > $OUT/sq_counters.txt 2>&1
fi
# keep the merge small: raw traces are not needed
rm -rf $OUT/kstats $OUT/pmc_fetch $OUT/pmc_write $OUT/sq1 $OUT/sq2
ls -la $OUT
