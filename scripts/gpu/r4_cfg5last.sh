#!/bin/bash
# r4, the round's last build: config 5 at 24 x 8 Mbp and at 24 x 40 Mbp (three in flight), without the CPU leg
out=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4_cfg5last}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
run() {  # name, extra args
  timeout 2400 python3 $GRAFT_REPO_ROOT/bench.py --config 5 --steps 4 --warmup 1 --e2e-steps 0 --no-cpu-baseline $2 > $out/$1.json 2> $out/$1.err
  echo "$1 rc $?"; tail -1 $out/$1.err | cut -c1-200
  python3 - $out/$1.json $1 <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    for k in ("value", "ms_per_step", "host_cpu_s_per_step", "setup_s", "device_memory_gb", "stage_ms"):
        print(sys.argv[2], k, d.get(k))
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
}
run cfg5_192 ""
run cfg5_960 "--backbone 40000000 --inflight 3"
