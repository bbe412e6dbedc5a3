#!/bin/bash
# r6: the lockstep kernel's sweep threshold (variants sw40, sw56 against the default 24), one batch in flight; and how many extensions the kernel hands on
cd $GRAFT_REPO_ROOT
AB_BENCH_ARGS="--inflight 1 --e2e-steps 0" bash scripts/gpu/ab.sh r6_sweep 2 prod sw40 sw56
cd /tmp && GC_DEBUG_TIMES=1 timeout 600 python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 2 --warmup 1 --inflight 1 --e2e-steps 0 2>&1 | grep "gc frag" | head -5
