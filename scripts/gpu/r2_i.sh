#!/bin/bash
# round 2, call I: chaining kernel v3
mkdir -p gpurun_out/r2i
timeout 2400 python -m pytest tests -m gpu -q -x > gpurun_out/r2i/pytest_all.log 2>&1; echo "rc $?" >> gpurun_out/r2i/pytest_all.log
tail -15 gpurun_out/r2i/pytest_all.log
timeout 500 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-long-pass > gpurun_out/r2i/bench_cfg2_nolong.json 2> gpurun_out/r2i/bench_cfg2_nolong.err
timeout 500 python bench.py --steps 8 --warmup 3 --no-cpu-baseline > gpurun_out/r2i/bench_cfg2.json 2> gpurun_out/r2i/bench_cfg2.err
timeout 500 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-long-pass --split-gap 18 > gpurun_out/r2i/bench_gap18_nolong.json 2> gpurun_out/r2i/bench_gap18_nolong.err
for f in gpurun_out/r2i/*.json; do echo $f; python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(d["value"], d["ms_per_step"], d["stage_ms"], d["roofline"]["frac"])
except Exception as e: print("ERR", e)
PY
done
