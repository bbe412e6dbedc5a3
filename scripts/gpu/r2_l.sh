#!/bin/bash
# timelines: one batch in flight with host stage times; two batches in flight under the kernel trace
out=$GRAFT_REPO_ROOT/gpurun_out/r2l
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
GC_DEBUG_TIMES=1 timeout 600 python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 4 --warmup 2 > $out/bench_1.json 2> $out/times_1.err
GC_DEBUG_TIMES=1 timeout 600 python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 8 --warmup 2 --inflight 2 > $out/bench_2.json 2> $out/times_2.err
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $out/trace2 -o t -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 6 --warmup 2 --inflight 2 > $out/trace2.log 2>&1
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $out/trace1 -o t -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 4 --warmup 2 > $out/trace1.log 2>&1
# keep only the kernel trace csv, compressed
for d in trace1 trace2; do f=$(find $out/$d -name "*kernel_trace.csv" | head -1); cut -d, -f4,5,7,8,10,11 "$f" | gzip > $out/$d.csv.gz; rm -rf $out/$d; done
ls -la $out
