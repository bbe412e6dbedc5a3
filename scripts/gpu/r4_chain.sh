#!/bin/bash
# r4: k_chain's scratch launch with the entries grouped by component: the chaining tests, config 5 at 24 x 8 Mbp and at 24 x 40 Mbp (three in flight)
out=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4_chain}
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 900 python3 -m pytest tests -q -m gpu -x -k "chain or stitch or long_reads or tangle or batches_in_flight" 2>&1 | tail -3 | tee $out/pytest.txt
cd /tmp && export TMPDIR=/tmp
run() {  # name, extra args
  timeout 2400 python3 $GRAFT_REPO_ROOT/bench.py --config 5 --steps 4 --warmup 1 --e2e-steps 0 --no-cpu-baseline $2 > $out/$1.json 2> $out/$1.err
  echo "$1 rc $?"; tail -2 $out/$1.err | cut -c1-300
  python3 - $out/$1.json $1 <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    for k in ("value", "ms_per_step", "host_cpu_s_per_step", "setup_s", "host_peak_rss_gb", "device_memory_gb", "inflight_for_device_memory", "stage_ms", "long_pass"):
        print(sys.argv[2], k, d.get(k))
    print(sys.argv[2], d["config"]["workload"], "in flight", d["config"]["batches_in_flight_per_gpu"])
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
}
run cfg5_192 ""
run cfg5_960 "--backbone 40000000 --inflight 3"
