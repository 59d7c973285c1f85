#!/bin/bash
# quick kernel-trace of cfg 3 at S = 1024 (per-kernel average durations), for iterating on the ML stage
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3q
rm -rf $O && mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats3 -o cfg3 -- python3 bench.py --config 3 --no-s1 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
python3 - <<'PY'
import csv, glob
for f in glob.glob('gpurun_out/r3q/stats3/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        print(r['Name'][:70].ljust(70), r['Calls'], 'avg_us', round(float(r['AverageNs'])/1e3,1), 'max_us', round(float(r['MaxNs'])/1e3,1))
PY
find $O -name "*kernel_trace.csv" -delete
