#!/bin/bash
# r4: the whole-read token taken at the first extension launch and dropped at the last count (production) against the token around the whole pass (GC_LONG_TOKEN_EARLY=1)
out=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4_lazy}
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 1500 python3 -m pytest tests -q -m gpu -x 2>&1 | tail -5 | tee $out/pytest.txt
cd /tmp && export TMPDIR=/tmp
run() {  # name, early, inflight
  env GC_LONG_TOKEN_EARLY=$2 timeout 600 python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --e2e-steps 0 --sv-leg-steps 0 --repeats-leg-steps 0 --steps 10 --warmup 4 --inflight $3 > $out/$1.json 2> $out/$1.err
  python3 - $out/$1.json $1 <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], d["value"], d["ms_per_step"], "long", d["stage_ms"]["k_long_extend_all_rounds"], d["stage_ms"]["whole_read_pass_wall"], "mem", d.get("device_memory_gb", {}).get("in_use_after_timed_steps"))
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
}
for r in 1 2 3; do
  run lazy5_$r 0 5
  run early5_$r 1 5
done
run lazy6_1 0 6
run lazy4_1 0 4
