#!/bin/bash
# column store in the whole-read kernel: parity tests, then A/B bench -> gpurun_out/$1
out=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $out
export GC_LONG_SM=0
timeout 1500 python -m pytest tests -q -m gpu -x > $out/pytest.log 2>&1
tail -5 $out/pytest.log
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift
  env "$@" timeout 600 python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 8 --warmup 2 > $out/bench_$name.json 2> $out/bench_$name.err
  python3 -c "
import json
try:
    d=json.loads(open('$out/bench_$name.json').read().strip().splitlines()[-1]); print('$name', d['value'], d['ms_per_step'], 'long', d['stage_ms']['k_long_extend_all_rounds'], 'wall', d['stage_ms']['whole_read_pass_wall'], 'reruns', d['long_pass']['plain_layout_reruns'], 'frac', d['roofline']['frac'])
except Exception as e: print('$name failed', e)
"
  tail -2 $out/bench_$name.err; }
run nocols GC_LONG_MAX_COLS=0
run cols X=1
run nocols1 GC_LONG_MAX_COLS=0 GC_BENCH_INFLIGHT=1
run cols1 GC_BENCH_INFLIGHT=1
run cols_hwq4 GPU_MAX_HW_QUEUES=4
