#!/bin/bash
# SQ counters of conv1x1_pipe launches on selected shapes (tools/g1_sweep.py filter), three rocprofv3 --pmc passes -> gpurun_out/$1/sq_g1.txt
#   gpurun -- 'bash tools/pmc_g1.sh r06_x "res4 2a"'
set -u
TAG=${1:-r06_pg}
SHAPE=${2:-"res4 2a"}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES -d $OUT/sq1 -o sq1 --output-format csv -- python $R/tools/g1_sweep.py "$SHAPE" > $OUT/sq1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $OUT/sq2 -o sq2 --output-format csv -- python $R/tools/g1_sweep.py "$SHAPE" > $OUT/sq2.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_INSTS_SMEM -d $OUT/sq3 -o sq3 --output-format csv -- python $R/tools/g1_sweep.py "$SHAPE" > $OUT/sq3.log 2>&1
cd $R
for f in $(find $OUT/sq1 $OUT/sq2 $OUT/sq3 -name "*counter_collection.csv"); do python tools/pmc_summary.py $f conv1x1_pipe; done > $OUT/sq_g1.txt 2>&1
rm -rf $OUT/sq1 $OUT/sq2 $OUT/sq3
tail -3 $OUT/sq1.log
