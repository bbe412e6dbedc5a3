#!/bin/bash
# round 3: state-machine whole-read kernel: parity tests, then bench variants -> gpurun_out/$1
out=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $out
timeout 1500 python -m pytest tests -q -m gpu -x -k "whole_read or batches_in_flight or config3 or config5 or long_reads or edge_cases_whole or chained_alignment" > $out/pytest_sm.log 2>&1
tail -8 $out/pytest_sm.log
cd /tmp && export TMPDIR=/tmp
run() { # name, env...
  name=$1; shift
  env "$@" timeout 600 python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 6 --warmup 2 > $out/bench_$name.json 2> $out/bench_$name.err
  python3 -c "
import json
try:
    d=json.loads(open('$out/bench_$name.json').read().strip().splitlines()[-1]); print('$name', d['value'], d['ms_per_step'], 'long', d['stage_ms']['k_long_extend_all_rounds'], 'wall', d['stage_ms']['whole_read_pass_wall'], 'rounds', d['long_pass']['rounds'], 'reruns', d['long_pass']['plain_layout_reruns'])
except Exception as e: print('$name failed', e)
"
  tail -2 $out/bench_$name.err
}
run old GC_LONG_SM=0
run sm_auto GC_LONG_SM=1
run sm_auto_1 GC_LONG_SM=1 GC_BENCH_INFLIGHT=1
run old_1 GC_LONG_SM=0 GC_BENCH_INFLIGHT=1
run sm8 GC_LONG_SM_LANES=8 GC_BENCH_INFLIGHT=1
run sm16 GC_LONG_SM_LANES=16 GC_BENCH_INFLIGHT=1
run sm32 GC_LONG_SM_LANES=32 GC_BENCH_INFLIGHT=1
run sm64 GC_LONG_SM_LANES=64 GC_BENCH_INFLIGHT=1
