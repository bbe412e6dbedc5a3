#!/bin/bash
# kernel timeline of the default bench (two batches in flight) -> gpurun_out/$1/timeline.txt
out=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $out/trace -o t -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 4 --warmup 2 > $out/trace.log 2>&1
f=$(find $out/trace -name "*kernel_trace.csv" | head -1)
python3 $GRAFT_REPO_ROOT/scripts/round_timeline.py $f 3 > $out/timeline.txt 2>&1
head -120 $out/timeline.txt
tail -3 $out/trace.log
rm -rf $out/trace
