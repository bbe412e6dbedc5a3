#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r2w
mkdir -p $out
timeout 1500 python -m pytest tests -q -m gpu -x > $out/pytest_all.log 2>&1
tail -3 $out/pytest_all.log
bash scripts/gpu/ab.sh r2w 3 notab prod
