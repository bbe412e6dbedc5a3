#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r2ac
mkdir -p $out
timeout 1500 python -m pytest tests -q -m gpu -x > $out/pytest_all.log 2>&1
tail -3 $out/pytest_all.log
AB_BENCH_ARGS="--inflight 1" bash scripts/gpu/ab.sh r2ac 3 nosplit prod
