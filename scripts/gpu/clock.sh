#!/bin/bash
# effective shader clock during the dominant kernel: GRBM_GUI_ACTIVE / 8 XCDs / dispatch duration (MI355X guide, DVFS section) over one cfg2 batch
out=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 90 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/clk -o p -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --e2e-steps 0 --sv-leg-steps 0 --repeats-leg-steps 0 --steps 1 --warmup 0 --inflight 1 > $out/clk.log 2>&1
python3 - $out <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
dur = {}
for f in glob.glob(out + "/clk/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Dispatch_Id"]] = (r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
acc = collections.defaultdict(lambda: [0.0, 0.0, 0])
for f in glob.glob(out + "/clk/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != "GRBM_GUI_ACTIVE" or r["Dispatch_Id"] not in dur:
            continue
        name, ns = dur[r["Dispatch_Id"]]
        name = name.split("(")[0].replace("void gcdev::", "").replace("gcdev::", "")
        if ns < 1_000_000:
            continue          # (the quotient reads high on short dispatches)
        a = acc[name[:30]]
        a[0] += float(r["Counter_Value"]); a[1] += ns; a[2] += 1
with open(out + "/clock.txt", "w") as o:
    for k, (cyc, ns, n) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:8]:
        line = f"{k:32s} dispatches {n:3d}  {ns / 1e6:8.1f} ms  effective clock {cyc / 8 / ns:.3f} GHz"
        print(line); o.write(line + "\n")
PY
rm -rf $out/clk
