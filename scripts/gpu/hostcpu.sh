#!/bin/bash
# host CPU seconds per step for different pool sizes, waits and batches in flight
out=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
run() { # name, env..., -- bench args
  name=$1; shift
  env "$@" timeout 600 python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 8 --warmup 2 $BARGS > $out/$name.json 2> $out/$name.err
  python3 -c "
import json
d=json.loads(open('$out/$name.json').read().strip().splitlines()[-1]); print('$name', d['value'], d['ms_per_step'], 'cpu s/step', d['host_cpu_s_per_step'], 'glue', d['stage_ms']['seed_glue_wall'], 'asm', d['stage_ms']['host_result_assembly'])"
}
BARGS="--inflight 2" run inf2_spin GC_SPIN_SYNC=1
BARGS="--inflight 2" run inf2_sleep GC_SPIN_SYNC=0
BARGS="--inflight 1" run inf1_spin GC_SPIN_SYNC=1
BARGS="--inflight 1" run inf1_sleep GC_SPIN_SYNC=0
BARGS="--inflight 1" run inf1_sleep_t8 GC_SPIN_SYNC=0 GC_HOST_THREADS=8
BARGS="--inflight 1" run inf1_sleep_t4 GC_SPIN_SYNC=0 GC_HOST_THREADS=4
