#!/bin/bash
# r4, the round's last build: the two-rank dry run of the launch path, then config 5 (24 x 8 Mbp) with the CPU leg and the end-to-end legs
out=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4_last}
mkdir -p $out
bash $GRAFT_REPO_ROOT/scripts/gpu/tworank.sh ${1:-r4_last}/two 2>&1 | tail -8 | cut -c1-700
cd /tmp && export TMPDIR=/tmp
timeout 2400 python3 $GRAFT_REPO_ROOT/bench.py --config 5 --steps 4 --warmup 1 --e2e-steps 4 > $out/cfg5_full.json 2> $out/cfg5_full.err
echo "cfg5 rc $?"; tail -2 $out/cfg5_full.err | cut -c1-300
python3 - $out/cfg5_full.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
for k in ("value", "ms_per_step", "host_cpu_s_per_step", "parity_check", "setup_s", "host_peak_rss_gb", "device_memory_gb", "stage_ms"):
    print(k, d.get(k))
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["single_thread_reads_per_s"])
print("e2e", {k: (v["reads_per_s"], v.get("gaf_check")) for k, v in d["e2e"].items() if isinstance(v, dict) and "reads_per_s" in v}, d["e2e"].get("gam_check"))
PY
