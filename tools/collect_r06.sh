#!/bin/bash
# Round-6 profile collection, two gpurun calls on one frozen build (csrc/ fingerprint):
#   gpurun -- 'bash tools/collect_r06.sh A r06'    evidence: kernel traces (two lanes / one lane), PMC traffic per launch (inference configs),
#                                                  per-kernel PMC summaries + kernel traces (training configs) -> gpurun_out/r06/
#   (copy gpurun_out/r06/* to profiles/ with the tag as prefix: `python tools/collect_r06.py install r06`)
#   gpurun -- 'bash tools/collect_r06.sh B r06'    the bench lines -- every line finds the evidence of ITS build in profiles/
set -u
PHASE=${1:-A}; TAG=${2:-r06}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
FP=$(cd $R && python -c "import bench; print(bench.build_fingerprint())")
sumstats() { python $R/profiles/summarize.py $(find $1 -name "*kernel_stats.csv" | head -1) $2 "$3" $FP >> $OUT/collect.log 2>&1; rm -rf $1; }
if [ "$PHASE" = "A" ]; then
  for c in sqdet_infer sqdetplus_infer sqdet_infer_384 sqdet_sample_b1; do
    P=$c; [ $c = sqdet_infer ] && P=""; PRE=${P:+${P}_}
    python $R/bench.py --config $c --no-cpu-baseline --layer-table $OUT/${PRE}layer_table.json > /dev/null 2>>$OUT/collect.log
    rocprofv3 --kernel-trace --stats -d $OUT/ks -o ks --output-format csv -- python $R/bench.py --config $c --no-cpu-baseline > /dev/null 2>&1
    sumstats $OUT/ks $OUT/${PRE}kernel_stats.txt "rocprofv3 --kernel-trace --stats -- python bench.py --config $c --no-cpu-baseline"
    SQDET_SERVE_LANES=1 rocprofv3 --kernel-trace --stats -d $OUT/ks1 -o ks --output-format csv -- python $R/bench.py --config $c --no-cpu-baseline > /dev/null 2>&1
    sumstats $OUT/ks1 $OUT/${PRE}kernel_stats_1lane.txt "SQDET_SERVE_LANES=1 rocprofv3 --kernel-trace --stats -- python bench.py --config $c --no-cpu-baseline"
    PMC_CONFIG=$c rocprofv3 --pmc FETCH_SIZE -d $OUT/pf -o fetch --output-format csv -- python $R/tools/pmc_forward.py > /dev/null 2>&1
    PMC_CONFIG=$c rocprofv3 --pmc WRITE_SIZE -d $OUT/pw -o write --output-format csv -- python $R/tools/pmc_forward.py > /dev/null 2>&1
    LAYERS=$(python -c "import json; print(','.join(l['layer'] for l in json.load(open('$OUT/${PRE}layer_table.json'))['layers']))")
    (cd $R && python profiles/pmc_traffic.py $(find $OUT/pf -name "*counter_collection.csv" | head -1) $(find $OUT/pw -name "*counter_collection.csv" | head -1) $OUT/${PRE}hbm_traffic_pmc.json "$LAYERS" $c >> $OUT/collect.log 2>&1)
    rm -rf $OUT/pf $OUT/pw
  done
  for c in sqdet_train_fp32 res50_train_fp16 sqdet_train_fp16; do
    rocprofv3 --kernel-trace --stats -d $OUT/ks -o ks --output-format csv -- python $R/bench.py --config $c --no-cpu-baseline --no-graph > /dev/null 2>&1
    sumstats $OUT/ks $OUT/kernel_stats_$c.txt "rocprofv3 --kernel-trace --stats -- python bench.py --config $c --no-cpu-baseline --no-graph"
    CSVS=""
    for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES"; do
      D=$OUT/pk_$(echo $grp | cut -d' ' -f1)
      rocprofv3 --pmc $grp -d $D -o pk --output-format csv -- python $R/bench.py --config $c --no-cpu-baseline --no-graph --steps 4 --warmup 2 > /dev/null 2>&1
      CSVS="$CSVS $(find $D -name "*counter_collection.csv" | head -1)"
    done
    (cd $R && python profiles/pmc_kernels.py $OUT/pmc_kernels_$c.json $c $CSVS >> $OUT/collect.log 2>&1)
    rm -rf $OUT/pk_*
  done
  # stand-alone tables: the fire modules' 1x1 convs (batch 32 and 128: events + copy column) and the deep 1x1 shapes A/B
  (cd $R && python tools/fire1x1_standalone.py > $OUT/fire_1x1_standalone.txt 2>>$OUT/collect.log; python tools/fire1x1_standalone.py --batch 128 >> $OUT/fire_1x1_standalone.txt 2>>$OUT/collect.log)
  (cd $R && python tools/ab_conv1x1_shapes.py > $OUT/conv1x1_shapes_ab.txt 2>>$OUT/collect.log)
  sed -i '/amdgpu.ids/d' $OUT/fire_1x1_standalone.txt $OUT/conv1x1_shapes_ab.txt
  ls -la $OUT; tail -30 $OUT/collect.log
else
  for c in sqdet_infer sqdet_infer_384 sqdet_sample_b1 sqdetplus_infer sqdet_train_fp32 res50_train_fp16 sqdet_train_fp16; do
    python $R/bench.py --config $c > $OUT/bench_$c.json 2> $OUT/bench_$c.err
  done
  python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_sqdet_infer_20steps.json 2>> $OUT/bench_sqdet_infer.err
  SQDET_SERVE_LANES=1 python $R/bench.py --no-cpu-baseline > $OUT/bench_sqdet_infer_1lane.json 2>> $OUT/bench_sqdet_infer.err
  python $R/tools/nextrows_bench.py > $OUT/nextrows_bench.json 2>/dev/null
  for f in $OUT/bench_*.json; do python -c "
import json,sys
d=json.loads(open('$f').readline()); r=d.get('roofline') or {}
print('%-40s %10s %s  ms %s  frac %s traffic %s dom %s' % ('$(basename $f)', d.get('value'), d.get('unit'), d.get('ms_per_step'), r.get('frac'), r.get('traffic'), (r.get('dominant_kernel') or {}).get('frac') if isinstance(r.get('dominant_kernel'), dict) else r.get('rocprof_frac')))"; done
fi
