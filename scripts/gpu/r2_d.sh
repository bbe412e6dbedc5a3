#!/bin/bash
# round 2, call D: whole GPU suite, the default bench line (config 2, with the all-core CPU leg), batches in flight, SV workload, config 3
mkdir -p gpurun_out/r2d
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r2d/pytest_all.log 2>&1; echo "rc $?" >> gpurun_out/r2d/pytest_all.log
tail -3 gpurun_out/r2d/pytest_all.log
timeout 600 python bench.py --steps 5 --warmup 2 > gpurun_out/r2d/bench_cfg2.json 2> gpurun_out/r2d/bench_cfg2.err
timeout 400 python bench.py --steps 6 --warmup 2 --inflight 2 --no-cpu-baseline > gpurun_out/r2d/bench_cfg2_inflight2.json 2> gpurun_out/r2d/bench_cfg2_inflight2.err
timeout 400 python bench.py --steps 6 --warmup 3 --inflight 3 --no-cpu-baseline > gpurun_out/r2d/bench_cfg2_inflight3.json 2> gpurun_out/r2d/bench_cfg2_inflight3.err
timeout 400 python bench.py --steps 4 --warmup 1 --sv-fraction 0.2 --no-cpu-baseline > gpurun_out/r2d/bench_cfg2_sv20.json 2> gpurun_out/r2d/bench_cfg2_sv20.err
GC_DEBUG_TIMES=1 timeout 400 python bench.py --steps 2 --warmup 1 --sv-fraction 0.2 --no-cpu-baseline > /dev/null 2> gpurun_out/r2d/times_sv20.err
timeout 900 python bench.py --config 3 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r2d/bench_cfg3.json 2> gpurun_out/r2d/bench_cfg3.err
for f in gpurun_out/r2d/*.json; do echo $f; python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(d["value"], d["ms_per_step"], d["stage_ms"], d["roofline"]["frac"], d["decision"], d.get("cpu_baseline"), d.get("reads_upload"))
except Exception as e: print("ERR", e)
PY
done
tail -5 gpurun_out/r2d/*.err
