#!/bin/bash
# config 5 with environment variants on one box, one set-up: bash scripts/gpu/cfg5_ab.sh <tag> <backbone bp per chromosome> <variant> ...
# AB_ENV_<variant>="K=V K=V" holds a variant's switches ("base": none), CFG5_ARGS adds bench flags (default: --inflight 5), CFG5_ROUNDS repeats the list
tag=$1; b=$2; shift 2
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
setup=/tmp/gcsetup_ab_$b
for r in $(seq 1 ${CFG5_ROUNDS:-1}); do
  for v in "$@"; do
    envvar="AB_ENV_$v"
    env ${!envvar} timeout 1500 python3 $GRAFT_REPO_ROOT/bench.py --config 5 --backbone $b --setup-dir $setup --steps ${CFG5_STEPS:-15} --warmup ${CFG5_WARMUP:-5} --e2e-steps 0 --no-cpu-baseline ${CFG5_ARGS:---inflight 5} > $out/${v}_$r.json 2> $out/${v}_$r.err
    python3 - $out/${v}_$r.json $v <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    s = d["stage_ms"]
    print(sys.argv[2], "reads/s", d["value"], "ms", d["ms_per_step"], "in flight", d["config"]["batches_in_flight_per_gpu"], "k_extend", s["k_extend"], "long", s["k_long_extend_all_rounds"], "pass wall", s["whole_read_pass_wall"], "glue", s["seed_glue_wall"], "frag wall", s["wall_extend_to_chain_and_copies"], "cpu", d["host_cpu_s_per_step"])
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
  done
done
rm -rf $setup
