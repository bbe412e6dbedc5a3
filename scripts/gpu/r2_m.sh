#!/bin/bash
# post-pass decision on the pass's thread, non-blocking streams, stream priority knob
out=$GRAFT_REPO_ROOT/gpurun_out/r2m
mkdir -p $out
timeout 1500 python -m pytest tests -q -m gpu -x > $out/pytest_all.log 2>&1
tail -3 $out/pytest_all.log
cd /tmp && export TMPDIR=/tmp
for mode in none frag long; do
  for inf in 1 2; do
    GC_STREAM_PRIORITY=$mode GC_DEBUG_TIMES=1 timeout 600 python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 6 --warmup 2 --inflight $inf > $out/bench_${mode}_$inf.json 2> $out/times_${mode}_$inf.err
  done
done
ls $out
