#!/bin/bash
# round 2, call H: new chaining kernel (entries per cover path in LDS), path kernel with LDS rings, cache fuzz-use test
mkdir -p gpurun_out/r2h
timeout 2400 python -m pytest tests -m gpu -q -x > gpurun_out/r2h/pytest_all.log 2>&1; echo "rc $?" >> gpurun_out/r2h/pytest_all.log
tail -15 gpurun_out/r2h/pytest_all.log
timeout 500 python bench.py --steps 8 --warmup 3 --no-cpu-baseline > gpurun_out/r2h/bench_cfg2.json 2> gpurun_out/r2h/bench_cfg2.err
timeout 500 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-long-pass > gpurun_out/r2h/bench_cfg2_nolong.json 2> gpurun_out/r2h/bench_cfg2_nolong.err
GC_DEBUG_TIMES=1 timeout 400 python bench.py --steps 3 --warmup 1 --sv-fraction 0.2 --no-cpu-baseline > gpurun_out/r2h/bench_cfg2_sv20.json 2> gpurun_out/r2h/bench_cfg2_sv20.err
timeout 900 python bench.py --config 3 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r2h/bench_cfg3.json 2> gpurun_out/r2h/bench_cfg3.err
for f in gpurun_out/r2h/*.json; do echo $f; python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(d["value"], d["ms_per_step"], d["stage_ms"], d["roofline"]["frac"], d["decision"])
except Exception as e: print("ERR", e)
PY
done
grep "chained alignment traces" gpurun_out/r2h/bench_cfg2_sv20.err | tail -2
