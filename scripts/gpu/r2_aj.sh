#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r2aj
mkdir -p $out
timeout 1500 python -m pytest tests -q -m gpu -x > $out/pytest_all.log 2>&1
tail -3 $out/pytest_all.log
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do
for ord in 0 1; do
  GC_EXT_ORDER=$ord timeout 600 python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 4 --warmup 1 --no-long-pass --inflight 1 > $out/nolong_ord${ord}_$rep.json 2> /dev/null
  GC_EXT_ORDER=$ord timeout 600 python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 8 --warmup 2 > $out/ord${ord}_$rep.json 2> /dev/null
  python3 -c "
import json
a=json.loads(open('$out/nolong_ord${ord}_$rep.json').read().strip().splitlines()[-1]); d=json.loads(open('$out/ord${ord}_$rep.json').read().strip().splitlines()[-1])
print('order $ord', 'k_extend alone', a['stage_ms']['k_extend'], '| full', d['value'], d['ms_per_step'], d['stage_ms']['k_extend'])"
done
done
