#!/bin/bash
# per-phase stamps of the state-machine whole-read kernel (profiling build) -> gpurun_out/$1
out=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
export GC_LIBRARY=$GRAFT_REPO_ROOT/graphchainer_amd/libgraphchainer_amd_smstamps.so
for lanes in ${SM_LANES_LIST:-8 64}; do
  GC_LONG_SM_LANES=$lanes GC_BENCH_INFLIGHT=1 timeout 600 python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 1 --warmup 1 $BENCH_ARGS > $out/stamps_$lanes.json 2> $out/stamps_$lanes.err
  echo "lanes $lanes"; grep "sm stamps" $out/stamps_$lanes.err | tail -5
  python3 -c "
import json
d=json.loads(open('$out/stamps_$lanes.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], 'long', d['stage_ms']['k_long_extend_all_rounds'])"
done
