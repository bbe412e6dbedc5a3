#!/bin/bash
# (this container) gpurun with retries while no GPU slot is free: bash scripts/gpu/retry.sh <timeout-s> '<command>'
t=$1; shift
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout $t -- "$@" > /tmp/gpurun_retry.$$ 2>&1
  rc=$?
  if grep -q "status=transient" /tmp/gpurun_retry.$$; then sleep 60; continue; fi
  cat /tmp/gpurun_retry.$$; rm -f /tmp/gpurun_retry.$$
  exit $rc
done
echo "gave up: no GPU slot"; exit 3
