#!/bin/bash
# lean scalar column loop: parity + bench
out=$GRAFT_REPO_ROOT/gpurun_out/r2n
mkdir -p $out
timeout 1500 python -m pytest tests -q -m gpu -x > $out/pytest_all.log 2>&1
tail -3 $out/pytest_all.log
cd /tmp && export TMPDIR=/tmp
timeout 600 python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 6 --warmup 2 > $out/bench_lean.json 2> $out/bench_lean.err
tail -c 1500 $out/bench_lean.json
