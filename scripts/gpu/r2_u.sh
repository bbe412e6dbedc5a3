#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r2u
mkdir -p $out
timeout 1500 python -m pytest tests -q -m gpu -x > $out/pytest_all.log 2>&1
tail -3 $out/pytest_all.log
bash scripts/gpu/times.sh r2u
python3 - <<'PY'
import json, os
d = json.loads(open(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r2u/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["stage_ms"])
PY
