#!/bin/bash
# the fragment extension kernels with ONE batch in flight (the whole-read pass of the same batch still runs beside them): kernel stats, counters and HBM traffic of k_extend*
# usage: bash scripts/gpu/frag_alone.sh <tag> [slab]      (slab: also the plain-layout kernel for everything, GC_EXTEND_SLAB=1)
cd $GRAFT_REPO_ROOT
tag=${1:-frag_alone}
BENCH_ARGS="--inflight 1" bash scripts/gpu/kstats.sh ${tag}_frag | grep -i "extend\|build_anchors\|stitch\|edit\|chain\|glue"
PMC_KERNELS="k_extend" BENCH_ARGS="--inflight 1" bash scripts/gpu/pmc.sh ${tag}_pmc_frag
if [ "$2" = slab ]; then
  GC_EXTEND_SLAB=1 BENCH_ARGS="--inflight 1" bash scripts/gpu/kstats.sh ${tag}_slab | grep -i "k_extend"
  GC_EXTEND_SLAB=1 PMC_KERNELS="k_extend" BENCH_ARGS="--inflight 1" bash scripts/gpu/pmc.sh ${tag}_pmc_slab | grep -i "VALU\|WAIT_ANY\|WAVE_CYCLES"
fi
out=$GRAFT_REPO_ROOT/gpurun_out/${tag}_traffic
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc_$c -o p -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --e2e-steps 0 --steps 1 --warmup 0 --inflight 1 > $out/pmc_$c.log 2>&1
done
python3 - $out <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
tot = collections.defaultdict(float)
for f in glob.glob(out + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].replace("void gcdev::", "").replace("gcdev::", "")
        if name.startswith("k_extend") or name.startswith("k_build_anchors"):
            tot[(name[:28], r["Counter_Name"])] += float(r["Counter_Value"])
with open(out + "/traffic.txt", "w") as o:
    for k in sorted(tot):
        line = f"{k[0]:30s} {k[1]:24s} {tot[k]:.4e}"
        print(line); o.write(line + "\n")
PY
rm -rf $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE
