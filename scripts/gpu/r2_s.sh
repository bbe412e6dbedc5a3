#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r2s
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do
for w in 0 7 6 5; do
  GC_LONG_WAVES_PER_SIMD=$w timeout 600 python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 5 --warmup 2 > $out/w${w}_$rep.json 2> $out/w${w}_$rep.err
  python3 - $out/w${w}_$rep.json w$w <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], d["value"], d["ms_per_step"], d["stage_ms"])
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
done
done
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $out/trace -o t -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 3 --warmup 1 > $out/trace.log 2>&1
python3 - $out <<'PY'
import csv, glob, gzip, sys
out = sys.argv[1]
f = glob.glob(out + "/trace/**/*kernel_trace.csv", recursive=True)[0]
with open(f) as src, gzip.open(out + "/trace.csv.gz", "wt") as dst:
    w = csv.writer(dst)
    for r in csv.DictReader(src):
        w.writerow([r["Stream_Id"], r["Kernel_Name"].split("(")[0][:60], r["Start_Timestamp"], r["End_Timestamp"]])
PY
rm -rf $out/trace
