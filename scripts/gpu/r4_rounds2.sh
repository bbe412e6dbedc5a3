#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4_rounds2}
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 1800 python -m pytest tests -q -m gpu -x 2>&1 | tail -15 | tee $out/pytest.txt
cd /tmp && export TMPDIR=/tmp
line() { python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'], d['stage_ms']['k_long_extend_all_rounds'], d['stage_ms']['whole_read_pass_wall'], d['long_pass']['extensions_per_step'], d['long_pass']['rounds'], d['host_cpu_s_per_step'])"; }
for rep in 1 2; do
for inf in 5 1; do
  GC_LONG_ROUNDS=0 timeout 600 python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --e2e-steps 0 --steps 8 --warmup 3 --inflight $inf 2>$out/r3loop_$inf.err | tee $out/r3loop_$inf.json | line r3loop_inflight$inf
  timeout 600 python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --e2e-steps 0 --steps 8 --warmup 3 --inflight $inf 2>$out/dev_$inf.err | tee $out/dev_$inf.json | line devrounds_inflight$inf
done
done
timeout 900 python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 5 > $out/bench.json 2> $out/bench.err; tail -3 $out/bench.err
python3 - $out/bench.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
for k in ("value", "ms_per_step", "host_cpu_s_per_step", "parity_check", "e2e", "sv_leg", "repeats_leg", "stage_ms"):
    print(k, d.get(k))
PY
