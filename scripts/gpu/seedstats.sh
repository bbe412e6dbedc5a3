#!/bin/bash
cd /tmp && export TMPDIR=/tmp
python3 - <<'PY'
import os, sys, numpy as np
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import graphchainer_amd as gca
from graphchainer_amd.synth import SynthGraph
sg = SynthGraph(5_000_000, seed=7)
p = "/tmp/g.gfa"; sg.write_gfa(p)
reads = sg.sample_reads(1000, 10000, seed=11)
g = gca.AlignmentGraph(p); s = gca.MinimizerSeeder(g)
out = gca.Aligner(g, s, long_pass=False).align_reads(reads)
slots = int(out["counters"][4]) // 2
print("extension pairs launched", slots, "seeds the reference extends", int(out["seeds_extended"].sum()), "anchors", len(out["anchor_x"]))
PY
