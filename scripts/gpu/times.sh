#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
GC_DEBUG_TIMES=1 timeout 600 python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 5 --warmup 2 "${@:2}" > $out/bench.json 2> $out/times.err
grep "gc times" $out/times.err | tail -24
