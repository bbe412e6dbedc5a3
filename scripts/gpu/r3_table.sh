#!/bin/bash
# the measurement table of DESIGN.md §11 (round 3) -> gpurun_out/$1
out=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift
  timeout 900 python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline "$@" > $out/$name.json 2> $out/$name.err
  python3 -c "
import json
try:
    d=json.loads(open('$out/$name.json').read().strip().splitlines()[-1]); print('$name', d['value'], d['ms_per_step'], 'long', d['stage_ms']['k_long_extend_all_rounds'], 'cpu', d['host_cpu_s_per_step'], d['decision'], d['e2e'], d['reads_upload']['reads_per_s_including_upload'])
except Exception as e: print('$name failed', e)
"; tail -1 $out/$name.err; }
run cfg2_5inflight --steps 30 --warmup 5 --e2e-steps 0
run cfg2_3inflight --steps 20 --warmup 5 --e2e-steps 0 --inflight 3
run cfg2_1inflight --steps 10 --warmup 2 --e2e-steps 0 --inflight 1
run cfg2_sv20 --steps 10 --warmup 3 --e2e-steps 2 --sv-fraction 0.2
run cfg3 --config 3 --steps 2 --warmup 1 --e2e-steps 0
