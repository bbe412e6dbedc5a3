#!/bin/bash
# rocprofv3 kernel stats of the default bench -> gpurun_out/$1/kernel_stats.csv
out=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o s -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 3 --warmup 2 --e2e-steps 0 $BENCH_ARGS > $out/stats.log 2>&1
f=$(find $out/stats -name "*kernel_stats.csv" | head -1)
cp $f $out/kernel_stats.csv
python3 - $out/kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
passes = 5
for r in rows[:32]:
    print(f"{r['Name'].replace('gcdev::','').replace('void ','').split('(')[0][:36]:36s} calls {int(r['Calls']):5d}  total/pass {float(r['TotalDurationNs'])/1e6/passes:8.2f} ms  avg {float(r['AverageNs'])/1e6:8.3f}  max {float(r['MaxNs'])/1e6:8.3f}  {r['Percentage']}%")
PY
tail -2 $out/stats.log
rm -rf $out/stats
