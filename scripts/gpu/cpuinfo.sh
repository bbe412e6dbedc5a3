#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/cpuinfo
mkdir -p $out
{
nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us /sys/fs/cgroup/cpu/cpu.cfs_period_us 2>/dev/null
python3 -c "import os; print(len(os.sched_getaffinity(0)), os.cpu_count())"
cat /sys/fs/cgroup/cpu.stat 2>/dev/null | head -8
cat /proc/self/cgroup | head -3
} > $out/info.txt 2>&1
cat $out/info.txt
cd /tmp && export TMPDIR=/tmp
for t in 8 16 32 96; do
  GC_HOST_THREADS=$t GC_DEBUG_TIMES=1 timeout 600 python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 5 --warmup 2 > $out/bench_t$t.json 2> $out/times_t$t.err
  echo "threads $t"; grep "seed expand" $out/times_t$t.err | tail -3; grep "timeline" $out/times_t$t.err | tail -3
  python3 -c "
import json,sys
d=json.loads(open('$out/bench_t$t.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['stage_ms']['seed_glue_wall'], d['stage_ms']['host_result_assembly'])"
done
cat /sys/fs/cgroup/cpu.stat 2>/dev/null | head -8
