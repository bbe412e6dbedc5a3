#!/bin/bash
# where the host CPU of a batch goes: bench.py under GC_DEBUG_TIMES (per-thread CPU per step, "[bench cpu]"; per-stage CPU of the batch threads and the worker pool, "[gc cpu]") with
# environment variants: bash scripts/gpu/hostcpu.sh <tag> <variant> ...   (AB_ENV_<variant>="K=V ..."; "base": none); HOSTCPU_ARGS adds bench flags
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  envvar="AB_ENV_$v"
  env ${!envvar} GC_DEBUG_TIMES=1 timeout 600 python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --e2e-steps 0 --steps 8 --warmup 5 $HOSTCPU_ARGS > $out/$v.json 2> $out/$v.err
  echo "== $v"
  grep "bench cpu" $out/$v.err | cut -c1-420
  grep "gc cpu\] main\|gc cpu\] worker" $out/$v.err | tail -2
  python3 - $out/$v.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("reads/s", d["value"], "ms", d["ms_per_step"], "cpu_s", d["host_cpu_s_per_step"])
except Exception as e:
    print("failed", e)
PY
done
