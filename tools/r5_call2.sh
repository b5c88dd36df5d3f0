#!/bin/bash
# GPU call 2 of round 5: read:write ceilings, stand-alone 1x1 sweep, SqueezeDet+ lane counts
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=gpurun_out/r05b
mkdir -p $O
hipcc --offload-arch=gfx950 -O3 -o /tmp/rw_ceiling tools/microbench/rw_ceiling.hip && timeout 120 /tmp/rw_ceiling > $O/rw_ceiling.txt 2>&1
cat $O/rw_ceiling.txt
timeout 300 python tools/c1_sweep.py > $O/c1_sweep.txt 2>&1
cat $O/c1_sweep.txt
for l in 2 3 4; do
  SQDET_SERVE_LANES=$l timeout 200 python bench.py --config sqdetplus_infer --no-cpu-baseline > $O/bench_sqdetplus_lanes$l.json 2> $O/bench_sqdetplus_lanes$l.err
  python -c "import json;d=json.load(open('$O/bench_sqdetplus_lanes$l.json'));print('sqdetplus lanes $l', d.get('value'), d.get('ms_per_step'), d.get('error'))"
done
