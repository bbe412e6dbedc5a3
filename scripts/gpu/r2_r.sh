#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r2r
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
GC_DEBUG_TIMES=1 timeout 600 python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 4 --warmup 2 > $out/bench_1.json 2> $out/times_1.err
grep "gc times" $out/times_1.err | tail -8
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o s -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 3 --warmup 1 > $out/stats.log 2>&1
f=$(find $out/stats -name "*kernel_stats.csv" | head -1); cp $f $out/kernel_stats.csv
f=$(find $out/stats -name "*kernel_trace.csv" | head -1); cut -d, -f4,5,7,8,10,11 "$f" | gzip > $out/trace.csv.gz; rm -rf $out/stats
# fragment pipeline alone (no whole-read pass): kernel durations without contention
GC_DEBUG_TIMES=1 timeout 600 python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 3 --warmup 1 --no-long-pass > $out/bench_nolong.json 2> $out/times_nolong.err
tail -c 900 $out/bench_nolong.json
