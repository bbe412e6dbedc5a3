#!/bin/bash
# what the driver runs at round end: the GPU tests, smoke(), and bench.py with its defaults (two runs: the first process on a fresh box is slower)
out=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $out
timeout 1500 python -m pytest tests -q -m gpu -x > $out/pytest_all.log 2>&1
tail -3 $out/pytest_all.log
# the same tests with every result array recycled and filled with 0xA5 before use: nothing may count on unwritten parts being zero
GC_RESULT_CACHE_MIN=1 GC_TEST_RESULT_CACHE_POISON=1 timeout 1500 python -m pytest tests -q -m gpu -x > $out/pytest_poison.log 2>&1
tail -2 $out/pytest_poison.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
timeout 900 python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 > $out/bench_a.json 2> $out/bench_a.err
timeout 900 python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 > $out/bench_b.json 2> $out/bench_b.err
for f in bench_a bench_b; do python3 -c "
import json
d=json.loads(open('$out/$f.json').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'], d['roofline']['frac'], d['cpu_baseline']['value'], d['cpu_baseline']['cores'], d['stage_ms']['k_long_extend_all_rounds'], d['parity_check']['mismatches'], {k: v.get('reads_per_s') for k, v in d['e2e'].items() if isinstance(v, dict) and 'reads_per_s' in v})"; done
# the multi-rank path on this one-GPU box (both ranks share the device: not a measurement)
bash $GRAFT_REPO_ROOT/scripts/gpu/tworank.sh $1_tworank 2>&1 | grep -v "^$" | cut -c1-700
