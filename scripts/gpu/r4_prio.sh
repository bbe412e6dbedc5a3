#!/bin/bash
# r4: s_setprio 3 in the latency-bound helper kernels (round loop's small kernels, seed glue, chaining, stitching) against the default, five batches in flight
out=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4_prio}
mkdir -p $out
AB_BENCH_ARGS="--e2e-steps 0 --sv-leg-steps 0 --repeats-leg-steps 0 --steps 10 --warmup 4" bash $GRAFT_REPO_ROOT/scripts/gpu/ab.sh ${1:-r4_prio}/ab 3 prod prio
cd /tmp && export TMPDIR=/tmp
export GC_LIBRARY=$GRAFT_REPO_ROOT/graphchainer_amd/libgraphchainer_amd_prio.so
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o s -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --e2e-steps 0 --sv-leg-steps 0 --repeats-leg-steps 0 --steps 3 --warmup 1 > $out/stats.log 2>&1
f=$(find $out/stats -name "*kernel_stats.csv" | head -1)
cp $f $out/kernel_stats_prio.csv; rm -rf $out/stats
python3 - $out/kernel_stats_prio.csv <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:12]:
    print(f"{r['Name'][:56]:56s} calls {r['Calls']:>4s} avg_ms {float(r['AverageNs'])/1e6:8.2f} {r['Percentage']}%")
PY
