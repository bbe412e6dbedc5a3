#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r2aa
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do
for mode in none long frag; do
  GC_STREAM_PRIORITY=$mode timeout 600 python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 8 --warmup 2 --inflight 2 > $out/${mode}_$rep.json 2> $out/${mode}_$rep.err
  python3 -c "
import json
d=json.loads(open('$out/${mode}_$rep.json').read().strip().splitlines()[-1]); print('$mode', d['value'], d['ms_per_step'], d['stage_ms']['k_long_extend_all_rounds'], d['stage_ms']['whole_read_pass_wall'])"
done
done
GC_DEBUG_TIMES=1 timeout 600 python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 6 --warmup 2 --inflight 2 > $out/times.json 2> $out/times.err
grep "timeline\|waited" $out/times.err | tail -8
