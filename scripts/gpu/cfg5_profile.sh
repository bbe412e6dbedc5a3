#!/bin/bash
# config 5 at one or more sizes: the bench line with the CPU leg and the parity sample (GC_DEBUG_TIMES: stage times and "[gc mem]" lines), then a rocprofv3 kernel trace of the same
# workload from the kept set-up (bench.py --setup-dir: the second process loads the index cache instead of building again).
# usage: bash scripts/gpu/cfg5_profile.sh <tag> [backbone bp per chromosome ...]      (8000000 = 192 Mbp, 40000000 = 960 Mbp); CFG5_ARGS adds bench flags; CFG5_ALONE=1 adds the trace with one batch in flight, CFG5_PMC=1 a counter pass
tag=$1; shift
sizes=${@:-8000000}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for b in $sizes; do
  setup=/tmp/gcsetup_$b
  GC_DEBUG_TIMES=1 timeout ${CFG5_TIMEOUT:-2400} python3 $GRAFT_REPO_ROOT/bench.py --config 5 --backbone $b --setup-dir $setup --steps ${CFG5_STEPS:-15} --warmup ${CFG5_WARMUP:-5} --e2e-steps ${CFG5_E2E:-0} $CFG5_ARGS > $out/cfg5_$b.json 2> $out/cfg5_$b.err
  echo "size $b rc $?"
  grep "gc mem" $out/cfg5_$b.err | tail -4
  python3 - $out/cfg5_$b.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("reads/s", d["value"], "ms", d["ms_per_step"], "in flight", d["config"]["batches_in_flight_per_gpu"], "parity", d["parity_check"] and (d["parity_check"]["read_results_compared"], d["parity_check"]["mismatches"]))
    print("stage_ms", d["stage_ms"]); print("mem", d["device_memory_gb"], d["inflight_for_device_memory"], "rss", d["host_peak_rss_gb"]); print("setup", d["setup_s"]); print("cpu", d["cpu_baseline"] and d["cpu_baseline"]["value"], "host cpu s", d["host_cpu_s_per_step"], "ties", d["unpinned_tie_reads"]["reads_whose_output_pass_had_a_tie_per_step"])
except Exception as e:
    print("no bench line:", e)
PY
  timeout 1500 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_$b -o s -- python3 $GRAFT_REPO_ROOT/bench.py --config 5 --backbone $b --setup-dir $setup --no-cpu-baseline --e2e-steps 0 --steps ${CFG5_TRACE_STEPS:-8} --warmup 5 $CFG5_ARGS > $out/stats_$b.log 2>&1   # (r6: a steady state - warm-up batches first)
  f=$(find $out/stats_$b -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $out/kernel_stats_$b.csv
  python3 - $out/kernel_stats_$b.csv <<'PY'
import csv, sys
try:
    rows = list(csv.DictReader(open(sys.argv[1])))
    for r in rows[:22]:
        print(f"{r['Name'].replace('gcdev::','').replace('void ','').split('(')[0][:40]:40s} calls {int(r['Calls']):5d}  total {float(r['TotalDurationNs'])/1e6:9.1f} ms  avg {float(r['AverageNs'])/1e6:8.3f}  max {float(r['MaxNs'])/1e6:8.3f}  {r['Percentage']}%")
except Exception as e:
    print("no kernel stats:", e)
PY
  rm -rf $out/stats_$b
  if [ -n "$CFG5_ALONE" ]; then   # the same trace with ONE batch in flight: what every kernel takes when nothing runs beside it
    timeout 1500 rocprofv3 --kernel-trace --stats --output-format csv -d $out/alone_$b -o s -- python3 $GRAFT_REPO_ROOT/bench.py --config 5 --backbone $b --setup-dir $setup --no-cpu-baseline --e2e-steps 0 --steps 2 --warmup 1 $CFG5_ARGS --inflight 1 > $out/alone_$b.log 2>&1
    f=$(find $out/alone_$b -name "*kernel_stats.csv" | head -1)
    [ -n "$f" ] && cp $f $out/kernel_stats_alone_$b.csv
    rm -rf $out/alone_$b
  fi
  if [ -n "$CFG5_PMC" ]; then
    timeout 1500 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_ANY --kernel-trace --output-format csv -d $out/pmc_$b -o p -- python3 $GRAFT_REPO_ROOT/bench.py --config 5 --backbone $b --setup-dir $setup --no-cpu-baseline --e2e-steps 0 --steps 1 --warmup 0 --inflight 1 $CFG5_ARGS > $out/pmc_$b.log 2>&1
    # r6: HBM traffic, one counter per pass
    for c in FETCH_SIZE WRITE_SIZE; do
      timeout 1500 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc_${b}_$c -o p -- python3 $GRAFT_REPO_ROOT/bench.py --config 5 --backbone $b --setup-dir $setup --no-cpu-baseline --e2e-steps 0 --steps 1 --warmup 0 --inflight 1 $CFG5_ARGS > $out/pmc_${b}_$c.log 2>&1
    done
    python3 - $out pmc_$b <<'PY'
import csv, glob, sys, collections
out, sub = sys.argv[1], sys.argv[2]
tot = collections.defaultdict(float)
for f in glob.glob(f"{out}/{sub}*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].replace("void gcdev::", "").replace("gcdev::", "")
        tot[(name[:34], r["Counter_Name"])] += float(r["Counter_Value"])
with open(f"{out}/{sub}_summary.txt", "w") as o:
    for k in sorted(tot):
        o.write(f"{k[0]:36s} {k[1]:22s} {tot[k]:.4e}\n")
print("pmc summary written")
PY
    rm -rf $out/pmc_$b $out/pmc_${b}_FETCH_SIZE $out/pmc_${b}_WRITE_SIZE
  fi
  rm -rf $setup
done
