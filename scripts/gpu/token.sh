#!/bin/bash
# whole-read token modes x batches in flight -> gpurun_out/$1
out=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
export GC_LONG_SM=0
for tok in 1 0 2; do for inflight in 2 3; do
  GC_LONG_TOKEN=$tok GC_BENCH_INFLIGHT=$inflight timeout 600 python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 10 --warmup 3 > $out/tok_${tok}_$inflight.json 2> $out/tok_${tok}_$inflight.err
  python3 -c "
import json
d=json.loads(open('$out/tok_${tok}_$inflight.json').read().strip().splitlines()[-1]); print('token $tok inflight $inflight', d['value'], d['ms_per_step'], 'long', d['stage_ms']['k_long_extend_all_rounds'], 'wall', d['stage_ms']['whole_read_pass_wall'], 'cpu', d['host_cpu_s_per_step'], d['stage_ms'])"
done; done
