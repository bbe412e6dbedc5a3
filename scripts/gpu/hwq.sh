#!/bin/bash
# effect of the number of hardware queues HIP streams are mapped onto (GPU_MAX_HW_QUEUES, default 4) -> gpurun_out/$1
out=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
export GC_LONG_SM=0
for q in ${HWQ_LIST:-4 8 16 32}; do
  for inflight in ${INFLIGHT_LIST:-2 3}; do
  GPU_MAX_HW_QUEUES=$q GC_BENCH_INFLIGHT=$inflight timeout 600 python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 15 --warmup 5 --e2e-steps 0 > $out/hwq_${q}_$inflight.json 2> $out/hwq_${q}_$inflight.err
  python3 -c "
import json
d=json.loads(open('$out/hwq_${q}_$inflight.json').read().strip().splitlines()[-1]); print('hwq $q inflight $inflight', d['value'], d['ms_per_step'], 'long', d['stage_ms']['k_long_extend_all_rounds'], 'wall', d['stage_ms']['whole_read_pass_wall'], 'cpu', d['host_cpu_s_per_step'])"
  done
done
