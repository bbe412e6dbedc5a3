#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r2z
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o s -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 3 --warmup 1 --no-long-pass > $out/stats.log 2>&1
f=$(find $out/stats -name "*kernel_stats.csv" | head -1); cp $f $out/kernel_stats_nolong.csv; rm -rf $out/stats
for inf in 1 2 3; do
  timeout 600 python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 8 --warmup 2 --inflight $inf > $out/inflight_$inf.json 2> $out/inflight_$inf.err
  python3 -c "
import json
d=json.loads(open('$out/inflight_$inf.json').read().strip().splitlines()[-1]); print('inflight $inf', d['value'], d['ms_per_step'], d['stage_ms']['k_long_extend_all_rounds'], d['stage_ms']['whole_read_pass_wall'])"
done
