#!/bin/bash
# GPU tests, then an A/B of the named variants against the production library (one batch in flight: the whole-read kernel's own time)
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
timeout 1500 python -m pytest tests -q -m gpu -x > $out/pytest_all.log 2>&1
tail -3 $out/pytest_all.log
AB_BENCH_ARGS="--inflight 1" bash scripts/gpu/ab.sh $tag 3 "$@" prod
