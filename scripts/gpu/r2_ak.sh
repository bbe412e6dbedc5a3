#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r2ak
mkdir -p $out
timeout 1500 python -m pytest tests -q -m gpu -x > $out/pytest_all.log 2>&1
tail -5 $out/pytest_all.log
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do
for lazy in 0 1; do
  GC_EXT_LAZY=$lazy timeout 600 python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 4 --warmup 1 --no-long-pass --inflight 1 > $out/nolong_lazy${lazy}_$rep.json 2> /dev/null
  GC_EXT_LAZY=$lazy timeout 600 python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 8 --warmup 2 > $out/lazy${lazy}_$rep.json 2> /dev/null
  python3 -c "
import json
a=json.loads(open('$out/nolong_lazy${lazy}_$rep.json').read().strip().splitlines()[-1]); d=json.loads(open('$out/lazy${lazy}_$rep.json').read().strip().splitlines()[-1])
print('lazy $lazy', 'alone: k_extend', a['stage_ms']['k_extend'], 'anchors', a['stage_ms']['k_build_anchors'], 'ext/step', a['extensions_per_step'], '| full', d['value'], d['ms_per_step'], d['stage_ms']['k_extend'], d['stage_ms']['k_build_anchors'])"
done
done
