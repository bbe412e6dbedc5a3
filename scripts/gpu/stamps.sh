#!/bin/bash
# section stamps of the whole-read kernel (make -C graphchainer_amd/csrc stamps first)
out=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
GC_LIBRARY=$GRAFT_REPO_ROOT/graphchainer_amd/libgraphchainer_amd_stamps.so python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 1 --warmup 1 2> $out/stamps.txt > $out/stamps.json
grep "gc stamps" $out/stamps.txt | tail -11
