#!/bin/bash
# r4: the lane-per-extension whole-read layout (GC_LONG_LANE=1, k_long_extend_lane) against k_long_extend<1>: parity, time alone and with five in flight, counters
out=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4_lane}
mkdir -p $out
cd $GRAFT_REPO_ROOT
GC_LONG_LANE=1 timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "whole_read or config5_shape or config3_shape or long_reads or reference_fixture" 2>&1 | tail -3 | tee $out/pytest_lane.txt
cd /tmp && export TMPDIR=/tmp
line() { python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'], d['stage_ms']['k_long_extend_all_rounds'], d['stage_ms']['whole_read_pass_wall'], d['roofline']['column_steps_per_s_G'], d['long_pass'])"; }
for inf in 1 5; do
  timeout 600 python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --e2e-steps 0 --steps 5 --warmup 2 --inflight $inf 2>$out/prod_$inf.err | tee $out/prod_$inf.json | line prod_$inf
  GC_LONG_LANE=1 timeout 900 python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --e2e-steps 0 --steps 5 --warmup 2 --inflight $inf 2>$out/lane_$inf.err | tee $out/lane_$inf.json | line lane_$inf
done
export GC_LONG_LANE=1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o s -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --e2e-steps 0 --steps 2 --warmup 1 --inflight 1 > $out/stats.log 2>&1
cp $(find $out/stats -name "*kernel_stats.csv" | head -1) $out/kernel_stats_lane.csv; rm -rf $out/stats
head -8 $out/kernel_stats_lane.csv
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d $out/pmc_a -o p -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --e2e-steps 0 --steps 1 --warmup 0 --inflight 1 > $out/pmc_a.log 2>&1
timeout 600 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INST_CYCLES_SALU --kernel-trace --output-format csv -d $out/pmc_b -o p -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --e2e-steps 0 --steps 1 --warmup 0 --inflight 1 > $out/pmc_b.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc_f -o p -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --e2e-steps 0 --steps 1 --warmup 0 --inflight 1 > $out/pmc_f.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/pmc_w -o p -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --e2e-steps 0 --steps 1 --warmup 0 --inflight 1 > $out/pmc_w.log 2>&1
python3 - $out <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
tot = collections.defaultdict(float)
for f in glob.glob(out + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].replace("void gcdev::", "").replace("gcdev::", "")
        if "k_long_extend" in name:
            tot[(name[:28], r["Counter_Name"])] += float(r["Counter_Value"])
with open(out + "/pmc_lane_summary.txt", "w") as o:
    for k in sorted(tot):
        line = f"{k[0]:30s} {k[1]:24s} {tot[k]:.4e}"
        print(line); o.write(line + "\n")
PY
rm -rf $out/pmc_a $out/pmc_b $out/pmc_f $out/pmc_w
