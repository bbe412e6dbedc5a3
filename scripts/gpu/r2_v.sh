#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r2v
mkdir -p $out
timeout 1500 python -m pytest tests -q -m gpu -x > $out/pytest_all.log 2>&1
tail -3 $out/pytest_all.log
cd /tmp && export TMPDIR=/tmp
timeout 900 python3 $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 2 > $out/bench.json 2> $out/bench.err
python3 - <<'PY'
import json, os
d = json.loads(open(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r2v/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["stage_ms"]); print(d["cpu_baseline"]); print(d["roofline"])
PY
