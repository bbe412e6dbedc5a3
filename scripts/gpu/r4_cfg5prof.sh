#!/bin/bash
# r4: kernel statistics of config 5 (24 x 8 Mbp, 2 000 x 50 kb reads, four in flight) with the round's last build
out=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4_cfg5prof}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 1500 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o s -- python3 $GRAFT_REPO_ROOT/bench.py --config 5 --no-cpu-baseline --e2e-steps 0 --steps 4 --warmup 1 > $out/stats.log 2>&1
f=$(find $out/stats -name "*kernel_stats.csv" | head -1)
cp $f $out/kernel_stats_cfg5.csv
head -16 $f | awk -F'","' '{printf "%s calls %s avg_ms %.3f total_ms %.1f pct %s\n", substr($1,2,64), $2, $4/1e6, $3/1e6, $5}'
tail -1 $out/stats.log | cut -c1-300
rm -rf $out/stats
