#!/bin/bash
# r4: all GPU tests, the default bench (CPU leg, e2e through the device encoder, sv and repeats legs), a per-round timeline of the whole-read pass with one batch in flight
out=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4_full}
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 1800 python -m pytest tests -q -m gpu -x 2>&1 | tail -5 | tee $out/pytest.txt
cd /tmp && export TMPDIR=/tmp
timeout 900 python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 5 > $out/bench.json 2> $out/bench.err; tail -3 $out/bench.err
python3 - $out/bench.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
for k in ("value", "ms_per_step", "host_cpu_s_per_step", "parity_check", "e2e", "sv_leg", "repeats_leg", "cpu_baseline", "stage_ms", "roofline"):
    print(k, d.get(k))
PY
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $out/trace -o t -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --e2e-steps 0 --steps 3 --warmup 1 --inflight 1 > $out/trace.log 2>&1
f=$(find $out/trace -name "*kernel_trace.csv" | head -1)
python3 $GRAFT_REPO_ROOT/scripts/round_timeline.py $f 2 > $out/timeline.txt 2>&1
grep -A60 "pass 2" $out/timeline.txt | head -80
rm -rf $out/trace
