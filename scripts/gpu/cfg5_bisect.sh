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
run base X=1
run nospec GC_LONG_SPECULATE=1
run nocols GC_LONG_MAX_COLS=0
run hostglue GC_DEVICE_GLUE=0
run nodiag GC_LIBRARY=$GRAFT_REPO_ROOT/graphchainer_amd/libgraphchainer_amd_nodiag.so
run nokinds GC_LIBRARY=$GRAFT_REPO_ROOT/graphchainer_amd/libgraphchainer_amd_nokinds.so
