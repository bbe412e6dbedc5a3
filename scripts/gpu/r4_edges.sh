#!/bin/bash
# r4: all GPU tests, then A/B of the whole-read kernel's per-tile changes (out-edges fetched together, 64-byte item store) against the builds without them
out=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4_edges}
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 1800 python -m pytest tests -q -m gpu 2>&1 | tail -4 | tee $out/pytest.txt
AB_BENCH_ARGS="--e2e-steps 0 --steps 8 --warmup 3" bash scripts/gpu/ab.sh ${1:-r4_edges}/ab 2 prod base noedges noitem
AB_BENCH_ARGS="--e2e-steps 0 --steps 6 --warmup 2 --inflight 1" bash scripts/gpu/ab.sh ${1:-r4_edges}/ab1 1 prod base noedges noitem
