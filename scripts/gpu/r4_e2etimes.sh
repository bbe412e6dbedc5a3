#!/bin/bash
# where an end-to-end batch spends its time: bench e2e legs with GC_DEBUG_TIMES
out=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4_e2etimes}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
GC_DEBUG_TIMES=1 timeout 900 python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 4 --warmup 2 --e2e-formats gaf --sv-leg-steps 0 --repeats-leg-steps 0 > $out/bench.json 2> $out/bench.err
python3 - $out/bench.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["e2e"])
PY
grep "output encoding on the device" $out/bench.err | tail -12
grep "gc_align_batch returned" $out/bench.err | tail -24 | awk '{print $(NF-1)}' | tr '\n' ' '; echo
grep "batch timeline" $out/bench.err | tail -8
