#!/bin/bash
# config 5, the bench line only (no kernel trace): bash scripts/gpu/cfg5_quick.sh <tag> <backbone bp per chromosome> [bench args...]
tag=$1; b=$2; shift 2
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
GC_DEBUG_TIMES=1 timeout ${CFG5_TIMEOUT:-2400} python3 $GRAFT_REPO_ROOT/bench.py --config 5 --backbone $b --steps ${CFG5_STEPS:-15} --warmup ${CFG5_WARMUP:-5} --e2e-steps ${CFG5_E2E:-0} "$@" > $out/cfg5_$b.json 2> $out/cfg5_$b.err
echo "size $b rc $?"
grep "gc mem" $out/cfg5_$b.err | tail -3
grep "gc ed" $out/cfg5_$b.err | head -300 > $out/ed_$b.txt
python3 - $out/cfg5_$b.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("reads/s", d["value"], "ms", d["ms_per_step"], "in flight", d["config"]["batches_in_flight_per_gpu"], "parity", d["parity_check"] and (d["parity_check"]["read_results_compared"], d["parity_check"]["mismatches"]))
    print("stage_ms", d["stage_ms"]); print("mem", d["device_memory_gb"], d["inflight_for_device_memory"], "rss", d["host_peak_rss_gb"]); print("setup", d["setup_s"]); print("cpu", d["cpu_baseline"] and d["cpu_baseline"]["value"], "host cpu s", d["host_cpu_s_per_step"])
    print("e2e", {k: (v.get("reads_per_s"), v.get("gaf_check")) if isinstance(v, dict) else v for k, v in (d.get("e2e") or {}).items()})
except Exception as e:
    print("no bench line:", e)
PY
