#!/bin/bash
# instruction / cycle counters of the whole-read kernel (two passes), summed per kernel
out=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS --kernel-trace --output-format csv -d $out/pmc_a -o p -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --e2e-steps 0 --steps 1 --warmup 0 $BENCH_ARGS > $out/pmc_a.log 2>&1
timeout 400 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR SQ_INST_CYCLES_SALU --kernel-trace --output-format csv -d $out/pmc_b -o p -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --e2e-steps 0 --steps 1 --warmup 0 $BENCH_ARGS > $out/pmc_b.log 2>&1
python3 - $out "${PMC_KERNELS:-k_long_extend<1 k_extend k_edit_distance<1}" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
wanted = sys.argv[2].split()
tot = collections.defaultdict(float)
for f in glob.glob(out + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].replace("void gcdev::", "").replace("gcdev::", "")
        if any(name.startswith(w) for w in wanted):
            tot[(name[:28], r["Counter_Name"])] += float(r["Counter_Value"])
with open(out + "/pmc_summary.txt", "w") as o:
    for k in sorted(tot):
        line = f"{k[0]:30s} {k[1]:24s} {tot[k]:.4e}"
        print(line); o.write(line + "\n")
PY
rm -rf $out/pmc_a $out/pmc_b
