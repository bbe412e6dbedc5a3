#!/bin/bash
# Round profile on the GPU box: bench line (with CPU baseline), rocprofv3 kernel stats, PMC passes, section stamps.
# Usage (from the repo root, through gpurun): bash scripts/profile_round.sh r01
set -u
tag=${1:-r01}
out=$GRAFT_REPO_ROOT/gpurun_out/profile_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 400 python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err
# (r6: a steady state - ten timed steps behind five warm-up steps; the r05 trace was --steps 3 --warmup 1, five of whose eight passes were the streams' lock-stepped first batches)
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o s -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --e2e-steps 0 --sv-leg-steps 0 --repeats-leg-steps 0 --steps 10 --warmup 5 > $out/stats.log 2>&1
timeout 400 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS --kernel-trace --output-format csv -d $out/pmc_a -o p -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --e2e-steps 0 --distinct-batches 1 --steps 1 --warmup 0 > $out/pmc_a.log 2>&1
timeout 400 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR SQ_INST_CYCLES_SALU --kernel-trace --output-format csv -d $out/pmc_b -o p -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --e2e-steps 0 --distinct-batches 1 --steps 1 --warmup 0 > $out/pmc_b.log 2>&1
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc_fetch -o p -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --e2e-steps 0 --distinct-batches 1 --steps 1 --warmup 0 > $out/pmc_fetch.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/pmc_write -o p -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --e2e-steps 0 --distinct-batches 1 --steps 1 --warmup 0 > $out/pmc_write.log 2>&1
GC_LIBRARY=$GRAFT_REPO_ROOT/graphchainer_amd/libgraphchainer_amd_stamps.so python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --e2e-steps 0 --steps 1 --warmup 1 2> $out/stamps.txt > /dev/null
# r6: the lockstep fragment kernel's sections (make -C graphchainer_amd/csrc variant NAME=fragstamps FLAGS=-DGC_FRAG_STAMPS)
GC_DEBUG_TIMES=1 GC_LIBRARY=$GRAFT_REPO_ROOT/graphchainer_amd/libgraphchainer_amd_fragstamps.so python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --e2e-steps 0 --sv-leg-steps 0 --repeats-leg-steps 0 --steps 5 --warmup 5 2>&1 > /dev/null | grep "gc frag" | tail -12 > $out/frag_stamps.txt
ls -R $out | head -40
