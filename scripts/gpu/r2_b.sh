#!/bin/bash
# round 2, call B: the alignment-path kernel and the chained-alignment output, then the whole GPU suite
mkdir -p gpurun_out/r2b
timeout 900 python -m pytest tests -m gpu -x -q -k "edit_path_kernel" > gpurun_out/r2b/pytest_path.log 2>&1; echo "rc $?" >> gpurun_out/r2b/pytest_path.log
tail -30 gpurun_out/r2b/pytest_path.log
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r2b/pytest_all.log 2>&1; echo "rc $?" >> gpurun_out/r2b/pytest_all.log
tail -40 gpurun_out/r2b/pytest_all.log
