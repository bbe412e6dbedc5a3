#!/bin/bash
# r4, the round's last build: config 3 (100 k reads, --colinear-split-gap 18) with its parity check, then the round profile (bench, kernel stats, PMC passes, stamps)
out=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4_wrap}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 1500 python3 $GRAFT_REPO_ROOT/bench.py --config 3 --steps 3 --warmup 1 --e2e-steps 0 --sv-leg-steps 0 --repeats-leg-steps 0 > $out/cfg3.json 2> $out/cfg3.err
echo "cfg3 rc $?"; tail -2 $out/cfg3.err | cut -c1-300
python3 - $out/cfg3.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
for k in ("value", "ms_per_step", "host_cpu_s_per_step", "parity_check", "stage_ms"):
    print(k, d.get(k))
print("cpu", d["cpu_baseline"]["value"], d["config"]["workload"])
PY
bash $GRAFT_REPO_ROOT/scripts/profile_round.sh r04g 2>&1 | tail -3
