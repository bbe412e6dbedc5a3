#!/bin/bash
# A/B on one box: bench.py with the production library and with named variants (graphchainer_amd/libgraphchainer_amd_<name>.so), interleaved.
# usage: bash scripts/gpu/ab.sh <outdir-tag> <rounds> <variant> [<variant> ...]      ("prod" = the production library)
tag=$1; rounds=$2; shift 2
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for r in $(seq 1 $rounds); do
  for v in "$@"; do
    libname=${v%%:*}; tag=${v##*:}           # "exp:split20" = library exp, environment AB_ENV_split20; a plain name is both
    lib=$GRAFT_REPO_ROOT/graphchainer_amd/libgraphchainer_amd_$libname.so
    if [ "$libname" = prod ]; then lib=$GRAFT_REPO_ROOT/graphchainer_amd/libgraphchainer_amd.so; fi
    envvar="AB_ENV_$tag"                     # AB_ENV_<tag>="K=V K=V": environment switches of that variant only
    env ${!envvar} GC_LIBRARY=$lib timeout 600 python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 5 --warmup 2 $AB_BENCH_ARGS > $out/${tag}_$r.json 2> $out/${tag}_$r.err
    python3 - $out/${tag}_$r.json $v <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], d["value"], d["ms_per_step"], "long", d["stage_ms"]["k_long_extend_all_rounds"], "wall", d["stage_ms"]["whole_read_pass_wall"], "k_extend", d["stage_ms"]["k_extend"], "anchors", d["stage_ms"]["k_build_anchors"], "parity", (d.get("parity_check") or {}).get("mismatches"))
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
  done
done
