#!/bin/bash
# config 5 on one GPU: a miniature first, then the sized run -> gpurun_out/$1
out=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 900 python3 $GRAFT_REPO_ROOT/bench.py --config 5 --chromosomes 4 --backbone 2000000 --reads 300 --batch 300 --steps 3 --warmup 1 --e2e-steps 1 > $out/cfg5_small.json 2> $out/cfg5_small.err
echo "small rc $?"; tail -c 1500 $out/cfg5_small.json; tail -3 $out/cfg5_small.err
if [ -n "$CFG5_FULL" ]; then
timeout ${CFG5_TIMEOUT:-2400} python3 $GRAFT_REPO_ROOT/bench.py --config 5 --steps 4 --warmup 1 --e2e-steps 1 $CFG5_ARGS > $out/cfg5_full.json 2> $out/cfg5_full.err
echo "full rc $?"; tail -c 2500 $out/cfg5_full.json; tail -3 $out/cfg5_full.err
fi
