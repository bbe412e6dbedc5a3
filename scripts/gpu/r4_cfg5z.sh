#!/bin/bash
# r4: config 5 at 24 x 40 Mbp (960 Mbp) with three batches in flight: the first build per stage, device and host memory
out=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4_cfg5z}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 2400 python3 $GRAFT_REPO_ROOT/bench.py --config 5 --steps 4 --warmup 1 --e2e-steps 0 --no-cpu-baseline --backbone 40000000 --inflight ${2:-3} > $out/cfg5_960.json 2> $out/cfg5_960.err
echo "rc $?"; tail -2 $out/cfg5_960.err | cut -c1-300
python3 - $out/cfg5_960.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
for k in ("value", "ms_per_step", "host_cpu_s_per_step", "setup_s", "host_peak_rss_gb", "device_memory_gb", "inflight_for_device_memory", "stage_ms", "long_pass"):
    print(k, d.get(k))
print(d["config"]["workload"], "in flight", d["config"]["batches_in_flight_per_gpu"])
PY
