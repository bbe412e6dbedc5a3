#!/bin/bash
# round 2, call C: config-shaped parity cases and the chaining kernel against the brute-force model
mkdir -p gpurun_out/r2c
timeout 1500 python -m pytest tests -m gpu -x -q -k "config5 or config3 or chain_kernel" > gpurun_out/r2c/pytest.log 2>&1; echo "rc $?" >> gpurun_out/r2c/pytest.log
tail -40 gpurun_out/r2c/pytest.log
