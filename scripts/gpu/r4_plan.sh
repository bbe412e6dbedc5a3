#!/bin/bash
# r4: speculation plans of the whole-read rounds (GC_LONG_PLAN) + first run of the device output encoder's tests
out=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4_plan}
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "gaf_output or json_and_gam or golden_files or chained_alignment_wins" 2>&1 | tail -15 | tee $out/pytest_output.txt
GC_LONG_PLAN=2,2,4,8 timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "whole_read or config5_shape or config3_shape or long_reads or reference_fixture or batches_in_flight" 2>&1 | tail -3 | tee $out/pytest_plan.txt
cd /tmp && export TMPDIR=/tmp
line() { python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'], d['stage_ms']['k_long_extend_all_rounds'], d['stage_ms']['whole_read_pass_wall'], d['long_pass']['extensions_per_step'], d['long_pass']['rounds'])"; }
for inf in 1 5; do
for plan in 1 2 2,2,2,8 2,2,4,8 2,4,8 3,3,8 2,3,8; do
  GC_LONG_PLAN=$plan timeout 600 python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --e2e-steps 0 --steps 6 --warmup 2 --inflight $inf 2>$out/plan_${plan}_$inf.err | tee $out/plan_${plan}_$inf.json | line plan_${plan}_inflight$inf
done
done
