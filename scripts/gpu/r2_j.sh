#!/bin/bash
# round 2, call J: path kernel without per-step global stores; SV workload; no-long-pass run (every read takes the chained alignment)
mkdir -p gpurun_out/r2j
timeout 2400 python -m pytest tests -m gpu -q -x -k "edit_path or chained or config5 or synthetic_parity" > gpurun_out/r2j/pytest.log 2>&1; echo "rc $?" >> gpurun_out/r2j/pytest.log
tail -5 gpurun_out/r2j/pytest.log
GC_DEBUG_TIMES=1 timeout 400 python bench.py --steps 3 --warmup 1 --sv-fraction 0.2 --no-cpu-baseline > gpurun_out/r2j/bench_cfg2_sv20.json 2> gpurun_out/r2j/bench_cfg2_sv20.err
GC_DEBUG_TIMES=1 timeout 500 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-long-pass > gpurun_out/r2j/bench_cfg2_nolong.json 2> gpurun_out/r2j/bench_cfg2_nolong.err
for f in gpurun_out/r2j/*.json; do echo $f; python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(d["value"], d["ms_per_step"], d["stage_ms"], d["decision"])
except Exception as e: print("ERR", e)
PY
done
grep "chained alignment traces" gpurun_out/r2j/*.err | tail -4
