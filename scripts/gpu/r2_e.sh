#!/bin/bash
# round 2, call E: suite with the overflow / shim tests and device-side read packing; bench with batches in flight (whole-read token)
mkdir -p gpurun_out/r2e
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r2e/pytest_all.log 2>&1; echo "rc $?" >> gpurun_out/r2e/pytest_all.log
tail -25 gpurun_out/r2e/pytest_all.log
for inf in 1 2 3; do
  timeout 500 python bench.py --steps 8 --warmup 3 --inflight $inf --no-cpu-baseline > gpurun_out/r2e/bench_cfg2_inflight$inf.json 2> gpurun_out/r2e/bench_cfg2_inflight$inf.err
done
GC_LONG_TOKEN=0 timeout 500 python bench.py --steps 8 --warmup 3 --inflight 2 --no-cpu-baseline > gpurun_out/r2e/bench_cfg2_inflight2_notoken.json 2> gpurun_out/r2e/bench_cfg2_inflight2_notoken.err
for f in gpurun_out/r2e/*.json; do echo $f; python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(d["value"], d["ms_per_step"], d["stage_ms"], d["roofline"]["frac"], d.get("reads_upload"))
except Exception as e: print("ERR", e)
PY
done
for f in gpurun_out/r2e/*.err; do echo $f; tail -n 3 $f; done
