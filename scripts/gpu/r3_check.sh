#!/bin/bash
# round 3: the GPU tests, smoke(), one default bench run (parity_check inside) -> gpurun_out/$1
out=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $out
timeout 2400 python -m pytest tests -q -m gpu -x ${PYTEST_ARGS} > $out/pytest_all.log 2>&1
tail -5 $out/pytest_all.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
timeout 900 python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 > $out/bench_a.json 2> $out/bench_a.err
echo "bench rc $?"
python3 -c "
import json
d=json.loads(open('$out/bench_a.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['cpu_baseline']['value'], d['parity_check'], d['host_cpu_s_per_step'], d['stage_ms'])"
tail -3 $out/bench_a.err
