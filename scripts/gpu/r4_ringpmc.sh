#!/bin/bash
# r4: HBM traffic and stand-alone time of k_extend with the backtrace column ring in LDS (production) and with the whole tile in the HBM slab (variant ring0)
out=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4_ringpmc}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
B="--no-cpu-baseline --e2e-steps 0 --sv-leg-steps 0 --repeats-leg-steps 0"
for v in prod ring0; do
  lib=$GRAFT_REPO_ROOT/graphchainer_amd/libgraphchainer_amd_$v.so
  if [ $v = prod ]; then lib=$GRAFT_REPO_ROOT/graphchainer_amd/libgraphchainer_amd.so; fi
  export GC_LIBRARY=$lib
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/${v}_$c -o p -- python3 $GRAFT_REPO_ROOT/bench.py $B --steps 1 --warmup 0 > $out/${v}_$c.log 2>&1
    f=$(find $out/${v}_$c -name "*counter_collection.csv" | head -1)
    python3 - $f $v $c <<'PY'
import csv, sys, collections
tot = collections.Counter(); calls = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"].split("(")[0]
    if r["Counter_Name"] == sys.argv[3]:
        tot[name] += float(r["Counter_Value"]); calls[name] += 1
for name in ("gcdev::k_extend", "gcdev::k_build_anchors"):
    print(sys.argv[2], sys.argv[3], name, "launches", calls[name], "KB", round(tot[name]), "GB", round(tot[name] / 1e6, 2))
PY
    rm -rf $out/${v}_$c
  done
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${v}_alone -o s -- python3 $GRAFT_REPO_ROOT/bench.py $B --steps 4 --warmup 1 --inflight 1 > $out/${v}_alone.log 2>&1
  f=$(find $out/${v}_alone -name "*kernel_stats.csv" | head -1)
  echo "== $v, one batch in flight"; grep "k_extend\|edit_distance" $f | awk -F'","' '{printf "%s calls %s avg_ms %.3f total_ms %.1f\n", substr($1,2,60), $2, $4/1e6, $3/1e6}'
  cp $f $out/kernel_stats_${v}_alone.csv; rm -rf $out/${v}_alone
done
