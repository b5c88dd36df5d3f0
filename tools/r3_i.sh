#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r03_i; mkdir -p $OUT; cd $R
run() { name=$1; shift; env "$@" python bench.py --no-cpu-baseline > $OUT/bench_$name.json 2>> $OUT/bench.err; python -c "
import json,sys; d=json.loads(open('$OUT/bench_$name.json').read().strip().splitlines()[-1]); r=d['roofline']; print('%-22s'%'$name', d['value'], d['ms_per_step'], 'fwd_only', d.get('forward_only_ms_per_step'), d.get('post_processing','')[:30], r['kernel'][:20], r['avg_launch_ms'], d['clocks']['before']['gfxclk_mhz'])"; }
run ride16 A=1
run ride8 SQDET_OPTIONS=dbg=308
run ride6 SQDET_OPTIONS=dbg=306
run ride12 SQDET_OPTIONS=dbg=312
run ride16b A=1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/ks1 -o ks --output-format csv -- python $R/bench.py --no-cpu-baseline > $OUT/ks1.log 2>&1
python $R/profiles/summarize.py $(find $OUT/ks1 -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_ride.txt "rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline (riders)" >> $OUT/ks1.log 2>&1
python - <<PY
import csv,glob
f=glob.glob("$OUT/ks1/**/*kernel_trace.csv", recursive=True)
if f:
    rows=list(csv.DictReader(open(f[0])))
    rows.sort(key=lambda r:int(r["Start_Timestamp"]))
    n=len(rows); sl=rows[n//2:n//2+30]
    t0=int(sl[0]["Start_Timestamp"])
    with open("$OUT/trace_slice_ride.txt","w") as o:
        for r in sl:
            o.write("%9.1f %9.1f %s grid=%s\n"%((int(r["Start_Timestamp"])-t0)/1e3,(int(r["End_Timestamp"])-t0)/1e3,r["Kernel_Name"][:60],r.get("Grid_Size_X")))
PY
rm -rf $OUT/ks1
head -18 $OUT/kernel_stats_ride.txt; cat $OUT/trace_slice_ride.txt
