#!/bin/bash
# round 2, call F: batches in flight after the memset change; lanes-per-wave variants with small tables / 4 waves per SIMD at 4x the reads
mkdir -p gpurun_out/r2f
for inf in 1 2; do
  GC_DEBUG_TIMES=0 timeout 500 python bench.py --steps 8 --warmup 3 --inflight $inf --no-cpu-baseline > gpurun_out/r2f/bench_cfg2_inflight$inf.json 2> gpurun_out/r2f/bench_cfg2_inflight$inf.err
done
export GC_LIBRARY=$PWD/graphchainer_amd/libgraphchainer_amd_wc16.so
for t in 1 4 8; do
  GC_LONG_TEAM=$t timeout 600 python bench.py --no-cpu-baseline --steps 2 --warmup 1 --reads 40000 --batch 40000 > gpurun_out/r2f/wc22_team_${t}_40k.json 2> gpurun_out/r2f/wc22_team_${t}_40k.err
done
unset GC_LIBRARY
for f in gpurun_out/r2f/*.json; do echo $f; python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(d["value"], d["ms_per_step"], d["stage_ms"], d["roofline"]["frac"])
except Exception as e: print("ERR", e)
PY
done
for f in gpurun_out/r2f/*.err; do echo $f; tail -n 3 $f; done
