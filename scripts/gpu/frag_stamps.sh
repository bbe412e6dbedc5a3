#!/bin/bash
# where the lockstep fragment kernel's waves spend their cycles (profiling build: make -C graphchainer_amd/csrc variant NAME=fragstamps FLAGS=-DGC_FRAG_STAMPS)
cd /tmp && GC_LIBRARY=$GRAFT_REPO_ROOT/graphchainer_amd/libgraphchainer_amd_fragstamps.so GC_DEBUG_TIMES=1 timeout 600 python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 2 --warmup 1 --inflight ${1:-1} --e2e-steps 0 2>&1 | grep "gc frag" | head -8
