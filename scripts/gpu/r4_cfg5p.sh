#!/bin/bash
# r4: config 5 at 24 x 40 Mbp (960 Mbp) WITH the CPU leg: the oracle builds the graph its own way and aligns the first reads, every timed batch is compared with it
out=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4_cfg5p}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 3300 python3 $GRAFT_REPO_ROOT/bench.py --config 5 --steps 4 --warmup 1 --e2e-steps 0 --backbone ${2:-40000000} --inflight ${3:-3} > $out/cfg5.json 2> $out/cfg5.err
echo "rc $?"; tail -2 $out/cfg5.err | cut -c1-300
python3 - $out/cfg5.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
for k in ("value", "ms_per_step", "host_cpu_s_per_step", "parity_check", "cpu_baseline", "setup_s", "host_peak_rss_gb", "device_memory_gb", "stage_ms"):
    print(k, d.get(k))
print(d["config"]["workload"], "in flight", d["config"]["batches_in_flight_per_gpu"])
PY
