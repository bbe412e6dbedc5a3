#!/bin/bash
# r4: GPU tests, then default vs device rounds vs large chain class, then the full bench
out=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4_ab4}
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 1800 python -m pytest tests -q -m gpu 2>&1 | tail -4 | tee $out/pytest.txt
GC_LONG_ROUNDS=1 timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "whole_read or config5_shape or config3_shape or long_reads or reference_fixture or batches_in_flight" 2>&1 | tail -2 | tee $out/pytest_devrounds.txt
cd /tmp && export TMPDIR=/tmp
line() { python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms']; print('$1', d['value'], d['ms_per_step'], 'long', s['k_long_extend_all_rounds'], s['whole_read_pass_wall'], 'ext', s['k_extend'], 'chain', s['k_chain'], 'rounds', d['long_pass']['rounds'], d['long_pass']['extensions_per_step'], 'cpu', d['host_cpu_s_per_step'])"; }
run() { name=$1; shift; env "$@" timeout 600 python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --e2e-steps 0 --steps 10 --warmup 3 2>$out/$name.err | tee $out/$name.json | line $name; }
for rep in 1 2; do
  run default_$rep GC_X=1
  run chainlarge_$rep GC_CHAIN_LARGE=1
done
timeout 900 python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err; tail -3 $out/bench.err
python3 - $out/bench.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
for k in ("value", "ms_per_step", "host_cpu_s_per_step", "parity_check", "e2e", "sv_leg", "repeats_leg", "stage_ms", "roofline"):
    print(k, d.get(k))
PY
