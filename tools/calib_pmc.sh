#!/bin/bash
# Box calibration with counters -> gpurun_out/$1/box_calibration.txt (copy to profiles/r05_box_calibration.txt):
#   (a) un-profiled: tools/calib_run.py (events + in-kernel s_memtime / s_memrealtime);
#   (b) rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES with --kernel-trace on the same script:
#       per dispatch the effective clock = GRBM_GUI_ACTIVE / 8 XCDs / kernel duration and the pipe's busy fraction.
#   gpurun -- 'bash tools/calib_pmc.sh r05_cal'
set -u
TAG=${1:-r05_cal}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $R/tools/calib_run.py --json $OUT/calib_unprofiled.json > $OUT/calib_unprofiled.txt 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES --kernel-trace -d $OUT/cal -o cal --output-format csv -- python $R/tools/calib_run.py --reps 1 > $OUT/calib_profiled.txt 2>&1
python $R/tools/calib_pmc_summary.py $(find $OUT/cal -name "*counter_collection.csv" | head -1) $(find $OUT/cal -name "*kernel_trace.csv" | head -1) $OUT/calib_unprofiled.txt > $OUT/box_calibration.txt 2>&1
rm -rf $OUT/cal
tail -40 $OUT/box_calibration.txt
