#!/bin/bash
# r4: the whole-read kernel (88 VGPRs, five waves per SIMD = 440 of a SIMD's 512 registers) capped at four / three waves per SIMD by an LDS pad (GC_LONG_WAVES_PER_SIMD): does the room it leaves
# let the helper kernels (k_stitch 87 VGPRs, k_extend 127) run beside it?
out=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4_cap}
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 600 python3 -m pytest tests -q -m gpu -x -k "seeding or seed or glue or whole_read or batches_in_flight" 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
run() {  # name, env
  env $2 timeout 600 python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --e2e-steps 0 --sv-leg-steps 0 --repeats-leg-steps 0 --steps 10 --warmup 4 > $out/$1.json 2> $out/$1.err
  python3 - $out/$1.json $1 <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], d["value"], d["ms_per_step"], "long", d["stage_ms"]["k_long_extend_all_rounds"], d["stage_ms"]["whole_read_pass_wall"], "glue", d["stage_ms"]["seed_glue_wall"], "chain..", d["stage_ms"]["wall_extend_to_chain_and_copies"])
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
}
for r in 1 2; do
  run five_$r GC_NOTHING=1
  run four_$r GC_LONG_WAVES_PER_SIMD=4
  run three_$r GC_LONG_WAVES_PER_SIMD=3
done
export GC_LONG_WAVES_PER_SIMD=4
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o s -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --e2e-steps 0 --sv-leg-steps 0 --repeats-leg-steps 0 --steps 3 --warmup 1 > $out/stats.log 2>&1
f=$(find $out/stats -name "*kernel_stats.csv" | head -1)
head -12 $f | awk -F'","' '{printf "%s calls %s avg_ms %.3f pct %s\n", substr($1,2,60), $2, $4/1e6, $5}'
cp $f $out/kernel_stats_four.csv; rm -rf $out/stats
