#!/bin/bash
# r4: k_edit_distance_block (a workgroup per pair for the pairs whose band covers most of the matrix): the edit-distance tests, cfg2 with and without it, config 5
out=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4_edblock}
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 1500 python3 -m pytest tests -q -m gpu -x 2>&1 | tail -3 | tee $out/pytest.txt
cd /tmp && export TMPDIR=/tmp
run() {  # name, env, args
  env $2 timeout 900 python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --e2e-steps 0 --sv-leg-steps 0 --repeats-leg-steps 0 $3 > $out/$1.json 2> $out/$1.err
  python3 - $out/$1.json $1 <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], d["value"], d["ms_per_step"], "long", d["stage_ms"]["k_long_extend_all_rounds"], d["stage_ms"]["whole_read_pass_wall"], "chain..", d["stage_ms"]["wall_extend_to_chain_and_copies"])
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
}
for r in 1 2; do
  run block_$r GC_ED_BLOCK=1 "--steps 10 --warmup 4"
  run noblock_$r GC_ED_BLOCK=0 "--steps 10 --warmup 4"
done
run cfg5_block GC_ED_BLOCK=1 "--config 5 --steps 4 --warmup 1"
run cfg5_noblock GC_ED_BLOCK=0 "--config 5 --steps 4 --warmup 1"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o s -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --e2e-steps 0 --sv-leg-steps 0 --repeats-leg-steps 0 --steps 4 --warmup 1 --inflight 1 > $out/stats.log 2>&1
f=$(find $out/stats -name "*kernel_stats.csv" | head -1)
grep "edit_distance" $f | awk -F'","' '{printf "%s calls %s avg_ms %.3f total_ms %.1f\n", substr($1,2,60), $2, $4/1e6, $3/1e6}'
cp $f $out/kernel_stats_alone.csv; rm -rf $out/stats
