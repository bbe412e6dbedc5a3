#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift
  env "$@" timeout 600 python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 10 --warmup 3 > $out/bench_$name.json 2> $out/bench_$name.err
  python3 -c "
import json
try:
    d=json.loads(open('$out/bench_$name.json').read().strip().splitlines()[-1]); print('$name', d['value'], d['ms_per_step'], 'long', d['stage_ms']['k_long_extend_all_rounds'], 'wall', d['stage_ms']['whole_read_pass_wall'], 'cpu', d['host_cpu_s_per_step'], 'glue', d['stage_ms']['seed_glue_wall'], 'k1wall', d['stage_ms']['wall_seed_lookup_and_copies'])
except Exception as e: print('$name failed', e)
"
  tail -2 $out/bench_$name.err; }
run base X=1
run prio_long GC_STREAM_PRIORITY=long
run prio_frag GC_STREAM_PRIORITY=frag
run hwq24 GPU_MAX_HW_QUEUES=24
run hwq12 GPU_MAX_HW_QUEUES=12
run inflight3 GC_BENCH_INFLIGHT=3
run inflight3_prio GC_BENCH_INFLIGHT=3 GC_STREAM_PRIORITY=long
run poll10 GC_SYNC_POLL_US=10
