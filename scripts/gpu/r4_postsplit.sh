#!/bin/bash
# r4: the build after gc_capi.hip was cut in two: all GPU tests (plain and with poisoned result arrays), smoke(), the default bench line
out=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4_postsplit}
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -3 | tee $out/pytest.txt
GC_RESULT_CACHE_MIN=1 GC_RESULT_CACHE_POISON=1 timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -2 | tee $out/pytest_poison.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
timeout 900 python3 $GRAFT_REPO_ROOT/bench.py > $out/bench.json 2> $out/bench.err
python3 -c "
import json
d=json.loads(open('$out/bench.json').read().strip().splitlines()[-1]); print('bench', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['cpu_baseline']['value'], d['stage_ms']['k_long_extend_all_rounds'], d['parity_check']['mismatches'], {k: v['reads_per_s'] for k, v in d['e2e'].items() if isinstance(v, dict) and 'reads_per_s' in v}, d['sv_leg']['reads_per_s'], d['repeats_leg']['reads_per_s'])"
