#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r2ae
mkdir -p $out
timeout 1500 python -m pytest tests -q -m gpu -x > $out/pytest_all.log 2>&1
tail -3 $out/pytest_all.log
cd /tmp && export TMPDIR=/tmp
for rep in 1 2 3; do
for inf in 1 2; do
  timeout 600 python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 8 --warmup 2 --inflight $inf > $out/inf${inf}_$rep.json 2> $out/inf${inf}_$rep.err
  python3 -c "
import json
d=json.loads(open('$out/inf${inf}_$rep.json').read().strip().splitlines()[-1]); print('inflight $inf', d['value'], d['ms_per_step'], d['stage_ms']['k_long_extend_all_rounds'], d['stage_ms']['whole_read_pass_wall'])"
done
done
