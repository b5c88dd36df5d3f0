#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r03_g; mkdir -p $OUT; cd $R
timeout 600 python -m pytest tests/test_gpu_model.py -m gpu -x -q -k "planted or deferred" > $OUT/pytest.txt 2>&1; echo "pytest rc=$?" >> $OUT/pytest.txt
tail -3 $OUT/pytest.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/ks1 -o ks --output-format csv -- python $R/bench.py --no-cpu-baseline > $OUT/ks1.log 2>&1
python $R/profiles/summarize.py $(find $OUT/ks1 -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_defer.txt "rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline (deferred post)" >> $OUT/ks1.log 2>&1
SQDET_POST_DEFER=0 rocprofv3 --kernel-trace --stats -d $OUT/ks2 -o ks --output-format csv -- python $R/bench.py --no-cpu-baseline > $OUT/ks2.log 2>&1
python $R/profiles/summarize.py $(find $OUT/ks2 -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_nodefer.txt "SQDET_POST_DEFER=0 rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline" >> $OUT/ks2.log 2>&1
# the kernel trace itself (start/end timestamps) of the deferred run: keep a slice for the overlap picture
python - <<PY
import csv,glob
f=glob.glob("$OUT/ks1/**/*kernel_trace.csv", recursive=True)
print(f)
if f:
    rows=list(csv.DictReader(open(f[0])))
    rows.sort(key=lambda r:int(r["Start_Timestamp"]))
    n=len(rows); sl=rows[n//2:n//2+60]
    t0=int(sl[0]["Start_Timestamp"])
    with open("$OUT/trace_slice_defer.txt","w") as o:
        for r in sl:
            o.write("%9.1f %9.1f %s q=%s wg=%s grid=%s\n"%((int(r["Start_Timestamp"])-t0)/1e3,(int(r["End_Timestamp"])-t0)/1e3,r["Kernel_Name"][:60],r.get("Queue_Id"),r.get("Workgroup_Size_X"),r.get("Grid_Size_X")))
PY
rm -rf $OUT/ks1 $OUT/ks2
head -22 $OUT/kernel_stats_defer.txt; cat $OUT/trace_slice_defer.txt | head -45
