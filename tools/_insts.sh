cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r03_insts; mkdir -p $OUT
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAVES SQ_BUSY_CYCLES -d $OUT/p1 -o p1 --output-format csv -- python $R/tools/pmc_forward.py > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_INSTS SQ_INSTS_BRANCH SQ_INSTS_VALU_CVT SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d $OUT/p2 -o p2 --output-format csv -- python $R/tools/pmc_forward.py > $OUT/p2.log 2>&1
cd $R
for f in $(find $OUT/p1 $OUT/p2 -name "*counter_collection.csv"); do python tools/pmc_summary.py $f sqdet; done > $OUT/insts.txt 2>&1
rm -rf $OUT/p1 $OUT/p2
grep -c . $OUT/insts.txt
