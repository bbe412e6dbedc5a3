#!/bin/bash
# r6: the whole-read kernel's column loop for nodes new in a slice without the two forcing operations (variant kind2) - parity tests with it, then an interleaved A/B
cd $GRAFT_REPO_ROOT
GC_LIBRARY=$GRAFT_REPO_ROOT/graphchainer_amd/libgraphchainer_amd_kind2.so timeout 1500 python -X faulthandler -m pytest tests -q -m gpu -x > gpurun_out/r6_kind2_pytest.txt 2>&1; echo "pytest (kind2) rc $?"; tail -2 gpurun_out/r6_kind2_pytest.txt
AB_BENCH_ARGS="--e2e-steps 0 --steps 10 --warmup 5" bash scripts/gpu/ab.sh r6_kind2 3 prod kind2
