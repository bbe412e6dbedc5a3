#!/bin/bash
# whole-read kernel capped at w waves per SIMD (an unused LDS allocation), five batches in flight -> gpurun_out/$1
out=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift
  env "$@" timeout 600 python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 20 --warmup 5 --e2e-steps 0 > $out/bench_$name.json 2> $out/bench_$name.err
  python3 -c "
import json
try:
    d=json.loads(open('$out/bench_$name.json').read().strip().splitlines()[-1]); print('$name', d['value'], d['ms_per_step'], 'long', d['stage_ms']['k_long_extend_all_rounds'], 'wall', d['stage_ms']['whole_read_pass_wall'], 'cpu', d['host_cpu_s_per_step'], 'fragwall', d['stage_ms']['wall_extend_to_chain_and_copies'])
except Exception as e: print('$name failed', e)
"
  tail -1 $out/bench_$name.err | cut -c1-200; }
run base X=1
run w7 GC_LONG_WAVES_PER_SIMD=7
run w6 GC_LONG_WAVES_PER_SIMD=6
run base2 X=1
run w5 GC_LONG_WAVES_PER_SIMD=5
