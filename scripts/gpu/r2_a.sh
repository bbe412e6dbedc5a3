#!/bin/bash
# round 2, call A: GPU tests on the current tree, baseline bench, and the lanes-per-wave sweep with 4x the reads in flight
mkdir -p gpurun_out/r2a
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r2a/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r2a/pytest.log
timeout 400 python bench.py --steps 5 --warmup 2 > gpurun_out/r2a/bench_base.json 2> gpurun_out/r2a/bench_base.err
for t in 1 4 16 64; do
  GC_LONG_TEAM=$t timeout 600 python bench.py --no-cpu-baseline --steps 2 --warmup 1 --reads 40000 > gpurun_out/r2a/team_${t}_40k.json 2> gpurun_out/r2a/team_${t}_40k.err
done
tail -2 gpurun_out/r2a/pytest.log
for f in gpurun_out/r2a/*.json; do echo $f; python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(d["value"], d["ms_per_step"], d["stage_ms"], d["roofline"]["frac"])
except Exception as e: print("ERR", e)
PY
done
