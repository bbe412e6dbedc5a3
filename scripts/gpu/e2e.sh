#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
GC_DEBUG_TIMES=1 timeout 900 python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 3 --warmup 2 --e2e-steps 3 > $out/e2e.json 2> $out/e2e.err
python3 -c "
import json
d=json.loads(open('$out/e2e.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['e2e'])"
grep "gc times\|gc cpu" $out/e2e.err | tail -30 | cut -c1-250
