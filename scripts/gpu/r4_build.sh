#!/bin/bash
# r4: first build of config 5's graph with the flat, threaded builder: the box's memory, then bench.py --config 5 (setup_s per stage) at the given backbone per chromosome
out=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4_build}
backbone=${2:-8000000}
mkdir -p $out
free -g | head -2 | tee $out/memory.txt; cat /sys/fs/cgroup/memory.max 2>/dev/null | tee -a $out/memory.txt; nproc | tee -a $out/memory.txt; cat /sys/fs/cgroup/cpu.max 2>/dev/null | tee -a $out/memory.txt
cd /tmp && export TMPDIR=/tmp
GC_DEBUG_TIMES=1 timeout 2400 python3 $GRAFT_REPO_ROOT/bench.py --config 5 --backbone $backbone --steps 3 --warmup 1 --e2e-steps 0 --no-cpu-baseline > $out/cfg5.json 2> $out/cfg5.err
echo "rc $?"; grep "gc build" $out/cfg5.err | grep -v "greedy\|shrink\|MPC index  " | head -40
python3 - $out/cfg5.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
for k in ("value", "ms_per_step", "setup_s", "host_peak_rss_gb", "device_memory_gb", "parity_check", "cpu_baseline", "config", "inflight_choice"):
    print(k, d.get(k))
PY
