#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $out
timeout 1500 python -m pytest tests -q -m gpu -x > $out/pytest_all.log 2>&1
tail -3 $out/pytest_all.log
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o s -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 3 --warmup 1 --inflight 1 > $out/stats.log 2>&1
f=$(find $out/stats -name "*kernel_stats.csv" | head -1); cp $f $out/kernel_stats.csv; rm -rf $out/stats
python3 - $out/kernel_stats.csv <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:12]:
    print(f"{r['Name'].split('(')[0][:50]:50s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e6:8.2f} ms")
PY
for i in 1 2 3; do timeout 600 python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 8 --warmup 2 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['decision'])"; done
