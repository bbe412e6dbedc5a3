#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o s -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 3 --warmup 1 --inflight 1 --no-long-pass > $out/stats.log 2>&1
f=$(find $out/stats -name "*kernel_stats.csv" | head -1); cp $f $out/kernel_stats.csv; rm -rf $out/stats
python3 - $out/kernel_stats.csv <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:12]:
    print(f"{r['Name'].split('(')[0][:50]:50s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e6:8.2f} ms")
PY
