#!/bin/bash
# r4: two-rank dry run (the in-flight choice, per-rank figures) and config 5 sized for one GPU with this round's build
out=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4_two_cfg5}
mkdir -p $out
bash $GRAFT_REPO_ROOT/scripts/gpu/tworank.sh ${1:-r4_two_cfg5}/two 2>&1 | tail -12
cd /tmp && export TMPDIR=/tmp
timeout 2400 python3 $GRAFT_REPO_ROOT/bench.py --config 5 --steps 4 --warmup 1 --e2e-steps 4 > $out/cfg5_full.json 2> $out/cfg5_full.err
echo "cfg5 rc $?"; tail -3 $out/cfg5_full.err
python3 - $out/cfg5_full.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
for k in ("value", "ms_per_step", "host_cpu_s_per_step", "parity_check", "e2e", "cpu_baseline", "setup_s", "device_memory_gb", "stage_ms", "long_pass"):
    print(k, d.get(k))
PY
