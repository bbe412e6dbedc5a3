#!/bin/bash
# usage: bash scripts/gpu/check.sh <tag> [bench args...]   - all GPU tests (whole log kept), then one default bench.py line (with the CPU leg unless told otherwise)
tag=${1:-check}; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 1800 python -X faulthandler -m pytest tests -q -m gpu -x $PYTEST_ARGS > $out/pytest.txt 2>&1
echo "pytest rc $?"
tail -15 $out/pytest.txt
cd /tmp && export TMPDIR=/tmp
timeout 900 python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 5 "$@" > $out/bench.json 2> $out/bench.err
echo "bench rc $?"
tail -3 $out/bench.err
python3 - $out/bench.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("reads/s", d["value"], "ms", d["ms_per_step"], "frac", d["roofline"]["frac"], "long", d["stage_ms"]["k_long_extend_all_rounds"], "cpu_s", d["host_cpu_s_per_step"])
    print("parity", d["parity_check"])
    print("ties", d["unpinned_tie_reads"])
    print("e2e", {k: (v.get("reads_per_s"), v.get("gaf_check")) if isinstance(v, dict) else v for k, v in (d.get("e2e") or {}).items()})
except Exception as e:
    print("no bench line:", e)
PY
