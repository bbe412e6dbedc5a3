#!/bin/bash
# round-end measurements: profile set, then the other workloads of DESIGN.md §11
bash scripts/profile_round.sh r02c
out=$GRAFT_REPO_ROOT/gpurun_out/r2final
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 900 python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 6 --warmup 2 > $out/cfg2.json 2> $out/cfg2.err
timeout 900 python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 6 --warmup 2 --inflight 1 > $out/cfg2_inflight1.json 2> $out/cfg2_inflight1.err
timeout 900 python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 4 --warmup 1 --sv-fraction 0.2 > $out/cfg2_sv20.json 2> $out/cfg2_sv20.err
timeout 1500 python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 2 --warmup 1 --config 3 > $out/cfg3.json 2> $out/cfg3.err
for f in cfg2 cfg2_inflight1 cfg2_sv20 cfg3; do python3 -c "
import json
d=json.loads(open('$out/$f.json').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'], d['roofline']['frac'], d['stage_ms']['k_long_extend_all_rounds'], d['decision'], d['reads_upload'])"; done
