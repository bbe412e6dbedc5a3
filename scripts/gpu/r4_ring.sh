#!/bin/bash
# r4: the backtrace column ring of k_extend (LDS, 8 / 16 columns) against the whole tile in the HBM slab, and the three-pairs-per-wave edit distance class on / off:
# GPU tests of the touched paths, interleaved bench runs, kernel stats of the production build with the class on and off
out=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4_ring}
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 1500 python3 -m pytest tests -q -m gpu -x 2>&1 | tail -5 | tee $out/pytest.txt
cd /tmp && export TMPDIR=/tmp
run() {  # name, library, extra env
  env GC_LIBRARY=$GRAFT_REPO_ROOT/graphchainer_amd/$2 $3 timeout 600 python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --e2e-steps 0 --sv-leg-steps 0 --repeats-leg-steps 0 --steps 8 --warmup 3 > $out/$1.json 2> $out/$1.err
  python3 - $out/$1.json $1 <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], d["value"], d["ms_per_step"], "k_extend", d["stage_ms"]["k_extend"], "long", d["stage_ms"]["k_long_extend_all_rounds"], d["stage_ms"]["whole_read_pass_wall"], "parity", d.get("parity_check"))
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
}
for r in 1 2; do
  run prod_$r libgraphchainer_amd.so GC_NOTHING=1
  run ring0_$r libgraphchainer_amd_ring0.so GC_NOTHING=1
  run ring16_$r libgraphchainer_amd_ring16.so GC_NOTHING=1
  run nothird_$r libgraphchainer_amd.so GC_ED_THIRD=0
done
for v in on off; do
  if [ $v = off ]; then export GC_ED_THIRD=0; fi
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_$v -o s -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --e2e-steps 0 --sv-leg-steps 0 --repeats-leg-steps 0 --steps 4 --warmup 2 > $out/stats_$v.log 2>&1
  f=$(find $out/stats_$v -name "*kernel_stats.csv" | head -1)
  echo "== kernel stats, three-pairs class $v"; grep -i "edit_distance\|k_extend\|pathseq" $f | awk -F'","' '{printf "%s calls %s avg_ms %.3f\n", substr($1,2,70), $2, $4/1e6}'
  cp $f $out/kernel_stats_$v.csv; rm -rf $out/stats_$v
done
