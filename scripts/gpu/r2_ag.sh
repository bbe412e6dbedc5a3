#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r2ag
mkdir -p $out
timeout 1500 python -m pytest tests -q -m gpu -x > $out/pytest_all.log 2>&1
tail -3 $out/pytest_all.log
cd /tmp && export TMPDIR=/tmp
usage() { grep -E "usage_usec|nr_throttled|throttled_usec" /sys/fs/cgroup/cpu.stat | tr '\n' ' '; echo; }
for rep in 1 2; do
for spin in 1 0; do
  usage
  GC_SPIN_SYNC=$spin timeout 600 python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 10 --warmup 2 > $out/spin${spin}_$rep.json 2> $out/spin${spin}_$rep.err
  python3 -c "
import json
d=json.loads(open('$out/spin${spin}_$rep.json').read().strip().splitlines()[-1]); print('spin $spin', d['value'], d['ms_per_step'], d['stage_ms']['k_long_extend_all_rounds'], d['stage_ms']['whole_read_pass_wall'], d['stage_ms']['host_seed_glue'])"
done
done
usage
