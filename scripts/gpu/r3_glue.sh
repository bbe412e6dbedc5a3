#!/bin/bash
# device seed glue: parity tests, then A/B bench -> gpurun_out/$1
out=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $out
timeout 1500 python -m pytest tests -q -m gpu -x > $out/pytest.log 2>&1
tail -8 $out/pytest.log
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift
  env "$@" timeout 600 python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 10 --warmup 3 > $out/bench_$name.json 2> $out/bench_$name.err
  python3 -c "
import json
try:
    d=json.loads(open('$out/bench_$name.json').read().strip().splitlines()[-1]); print('$name', d['value'], d['ms_per_step'], 'long', d['stage_ms']['k_long_extend_all_rounds'], 'wall', d['stage_ms']['whole_read_pass_wall'], 'cpu', d['host_cpu_s_per_step'], 'glue', d['stage_ms']['host_seed_glue'], 'k1wall', d['stage_ms']['wall_seed_lookup_and_copies'])
except Exception as e: print('$name failed', e)
"
  tail -2 $out/bench_$name.err; }
run hostglue GC_DEVICE_GLUE=0
run devglue X=1
run devglue3 GC_BENCH_INFLIGHT=3
run devglue_spin GC_SPIN_SYNC=1
run hostglue_spin GC_DEVICE_GLUE=0 GC_SPIN_SYNC=1
run devglue1 GC_BENCH_INFLIGHT=1
