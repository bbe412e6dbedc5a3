#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift
  env "$@" timeout 600 python3 $GRAFT_REPO_ROOT/bench.py --config 5 --chromosomes 4 --backbone 2000000 --reads 300 --batch 300 --steps 1 --warmup 1 --e2e-steps 0 --inflight 1 > $out/b_$name.json 2> $out/b_$name.err
  python3 -c "
import json
d=json.loads(open('$out/b_$name.json').read().strip().splitlines()[-1]); print('$name', d['parity_check']['mismatches'], d['parity_check'].get('fields_with_mismatches'), d['long_pass'])"
}
run cells24 GC_LONG_CELLS_PER_BASE=24
run cells48 GC_LONG_CELLS_PER_BASE=48
