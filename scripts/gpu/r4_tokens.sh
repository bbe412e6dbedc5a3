#!/bin/bash
# r4: two whole-read tokens per device (GC_LONG_TOKENS=2: two passes side by side, a scratch each) against one, at 4 / 5 / 6 batches in flight
out=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4_tokens}
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 900 python3 -m pytest tests -q -m gpu -x -k "batches_in_flight or edit_distance" 2>&1 | tail -5 | tee $out/pytest.txt
cd /tmp && export TMPDIR=/tmp
run() {  # name, tokens, inflight
  env GC_LONG_TOKENS=$2 timeout 600 python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --e2e-steps 0 --sv-leg-steps 0 --repeats-leg-steps 0 --steps 10 --warmup 4 --inflight $3 > $out/$1.json 2> $out/$1.err
  python3 - $out/$1.json $1 <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], d["value"], d["ms_per_step"], "long", d["stage_ms"]["k_long_extend_all_rounds"], d["stage_ms"]["whole_read_pass_wall"], "mem", d.get("device_memory_gb", {}).get("in_use_after_timed_steps"), "parity", (d.get("parity_check") or {}).get("mismatches"))
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
}
for r in 1 2; do
  run one5_$r 1 5
  run two5_$r 2 5
  run two6_$r 2 6
  run two4_$r 2 4
done
