#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r2ad
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
cat /sys/fs/cgroup/cpu.stat | grep -E "nr_throttled|throttled_usec"
GC_DEBUG_TIMES=1 timeout 900 python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 > $out/with_cpu.json 2> $out/with_cpu.err
cat /sys/fs/cgroup/cpu.stat | grep -E "nr_throttled|throttled_usec"
GC_DEBUG_TIMES=1 timeout 900 python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $out/no_cpu.json 2> $out/no_cpu.err
cat /sys/fs/cgroup/cpu.stat | grep -E "nr_throttled|throttled_usec"
for f in with_cpu no_cpu; do python3 -c "
import json
d=json.loads(open('$out/$f.json').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'], d['stage_ms']['host_seed_glue'], d['stage_ms']['k_long_extend_all_rounds'], d['setup_s']['graph_build_upload'])"; done
grep "seed expand" $out/with_cpu.err | tail -4
grep "seed expand" $out/no_cpu.err | tail -4
