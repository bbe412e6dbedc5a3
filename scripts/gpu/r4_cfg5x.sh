#!/bin/bash
# r4: config 5 (2 000 x 50 kb CLR-like reads, 24 x 8 Mbp): whole-read passes side by side (GC_LONG_TOKENS=2) and larger batches, against the default
out=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4_cfg5x}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
run() {  # name, env, extra args
  env $2 timeout 1200 python3 $GRAFT_REPO_ROOT/bench.py --config 5 --steps 4 --warmup 1 --e2e-steps 0 --no-cpu-baseline $3 > $out/$1.json 2> $out/$1.err
  python3 - $out/$1.json $1 <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], d["value"], d["ms_per_step"], "long", d["stage_ms"]["k_long_extend_all_rounds"], d["stage_ms"]["whole_read_pass_wall"], "mem", d["device_memory_gb"]["in_use_after_timed_steps"], "inflight", d["config"]["batches_in_flight_per_gpu"], "batch", d["config"]["batch"], "setup", d["setup_s"])
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
}
run default GC_NOTHING=1 ""
run two_tokens GC_LONG_TOKENS=2 ""
run batch4000 GC_NOTHING=1 "--reads 4000 --batch 4000 --inflight 2"
run batch4000_two GC_LONG_TOKENS=2 "--reads 4000 --batch 4000 --inflight 2"
