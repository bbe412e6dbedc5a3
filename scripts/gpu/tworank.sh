#!/bin/bash
# dry run of the multi-rank path on a one-GPU box (both ranks share the device; not a measurement)
out=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
export GC_BENCH_ALLOW_SHARED_GPU=1
for mode in "" "--strong"; do
  timeout 900 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 $GRAFT_REPO_ROOT/bench.py --gpus 2 --steps 3 --warmup 1 --reads 2000 --backbone 5000000 $mode > $out/two$mode.json 2> $out/two$mode.err
  echo "rc $?"; tail -c 600 $out/two$mode.json; echo; grep -i "error\|Traceback" $out/two$mode.err | head -3
done
