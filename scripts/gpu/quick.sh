#!/bin/bash
# GPU tests and three default bench runs
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
for i in 1 2 3; do timeout 600 python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 8 --warmup 2 $QUICK_ARGS 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['decision'], d['stage_ms']['k_long_extend_all_rounds'])"; done
