#!/bin/bash
# device minimizer index build: parity + timing at cfg2 size
mkdir -p gpurun_out/r2k
timeout 900 python -m pytest tests/test_index_cache.py -q -m gpu -x > gpurun_out/r2k/pytest_index.log 2>&1
tail -5 gpurun_out/r2k/pytest_index.log
timeout 1200 python - > gpurun_out/r2k/build_times.txt 2>&1 <<'PY'
import os, time
os.environ["GC_DEBUG_TIMES"] = "1"
import graphchainer_amd as gca
from graphchainer_amd.synth import SynthGraph
import numpy as np
sg = SynthGraph(50_800_000, seed=1)
p = "/tmp/cfg2.gfa"
t = time.time(); sg.write_gfa(p); print("write gfa", time.time() - t, flush=True)
t = time.time(); g = gca.AlignmentGraph(p); print("graph total", time.time() - t, flush=True)
for mode in ("device", "host", "device"):
    if mode == "host": os.environ["GC_SEEDER_BUILD"] = "host"
    else: os.environ.pop("GC_SEEDER_BUILD", None)
    t = time.time(); s = gca.MinimizerSeeder(g, 15, 20); dt = time.time() - t
    a = s.array("positions")
    print(mode, "seeder", round(dt, 3), "s", len(a), int(a[:1000].sum()), int(s.array("maxcount")[0]), flush=True)
PY
cat gpurun_out/r2k/build_times.txt | tail -30
