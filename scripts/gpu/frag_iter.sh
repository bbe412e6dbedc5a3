#!/bin/bash
# one iteration on the fragment kernel - GPU tests of the fragment pipeline, section stamps, the kernels of one batch in flight
cd $GRAFT_REPO_ROOT
timeout 1500 python -X faulthandler -m pytest tests -q -m gpu -x -k "${R6_TESTS:-parity or fragment or lazy or fixture or shapes or pools}" > gpurun_out/${1:-frag_iter}_pytest.txt 2>&1; echo "pytest rc $?"; tail -2 gpurun_out/${1:-frag_iter}_pytest.txt
bash scripts/gpu/frag_stamps.sh 1 | tail -2
BENCH_ARGS="--inflight 1" bash scripts/gpu/kstats.sh ${1:-frag_iter}_frag | grep -i "k_extend\|build_anchors"
