#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r2af
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for rep in 1 2 3; do
for tok in 1 2; do
  GC_LONG_TOKEN=$tok timeout 600 python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 10 --warmup 2 --inflight 2 > $out/tok${tok}_$rep.json 2> $out/tok${tok}_$rep.err
  python3 -c "
import json
d=json.loads(open('$out/tok${tok}_$rep.json').read().strip().splitlines()[-1]); print('token $tok', d['value'], d['ms_per_step'], d['stage_ms']['k_long_extend_all_rounds'], d['stage_ms']['whole_read_pass_wall'])"
done
done
GC_LONG_TOKEN=2 timeout 600 python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 10 --warmup 3 --inflight 3 > $out/tok2_inf3.json 2> $out/tok2_inf3.err
python3 -c "
import json
d=json.loads(open('$out/tok2_inf3.json').read().strip().splitlines()[-1]); print('token 2 inflight 3', d['value'], d['ms_per_step'], d['stage_ms']['k_long_extend_all_rounds'], d['stage_ms']['whole_read_pass_wall'])"
