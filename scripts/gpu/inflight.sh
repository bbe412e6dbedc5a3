#!/bin/bash
# bench.py with 3..6 batches in flight (the whole-read scratch is shared per device) -> gpurun_out/$1
out=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for k in ${INFLIGHT_LIST:-3 4 5 6}; do
  timeout 900 python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps $((4*k)) --warmup $k --e2e-steps 0 --inflight $k > $out/inflight_$k.json 2> $out/inflight_$k.err
  python3 - $out/inflight_$k.json $k <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("inflight", sys.argv[2], d["value"], d["ms_per_step"], d["stage_ms"]["k_long_extend_all_rounds"], d["stage_ms"]["whole_read_pass_wall"], d["device_memory_gb"], d["host_cpu_s_per_step"])
except Exception as e:
    print("inflight", sys.argv[2], "failed", e)
PY
  tail -2 $out/inflight_$k.err | cut -c1-300
done
