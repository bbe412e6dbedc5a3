#!/usr/bin/env python3
"""Turns the output of scripts/profile_round.sh (gpurun_out/profile_<tag>/) into the committed summaries under profiles/.
Usage: python scripts/collect_profiles.py gpurun_out/profile_r01g r01"""
import collections
import csv
import json
import os
import shutil
import sys

src, prefix = sys.argv[1], sys.argv[2]
out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")


def short(name):
    return name.replace("void ", "").replace("gcdev::", "").split("(")[0]


def sums(path):
    agg, launches = collections.defaultdict(float), collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        k = short(r["Kernel_Name"])
        agg[(k, r["Counter_Name"])] += float(r["Counter_Value"])
        launches[k].add(r["Dispatch_Id"])
    return agg, {k: len(v) for k, v in launches.items()}


# bench line and kernel stats
line = open(os.path.join(src, "bench.json")).read().strip().splitlines()[-1]
json.loads(line)
open(os.path.join(out, f"{prefix}_bench_cfg2.json"), "w").write(line + "\n")
shutil.copy(os.path.join(src, "stats", "s_kernel_stats.csv"), os.path.join(out, f"{prefix}_kernel_stats_cfg2.csv"))

# SQ counters, all kernels and the dominant one
rows = {}
batches = 1
for sub in ("pmc_a", "pmc_b", "pmc_fetch", "pmc_write"):
    agg, launches = sums(os.path.join(src, sub, "p_counter_collection.csv"))
    rows.update(agg)
    batches = max(batches, launches.get("k_seed_probe", 1))   # one k_seed_probe launch per batch: an r6 pass over `--steps 1` holds the timed step AND the same-read-set leg
header = (f"# rocprofv3 --pmc passes over: python3 bench.py --no-cpu-baseline --steps 1 --warmup 0  (cfg2, 10k x 10 kb reads; values summed over all launches of the pass = {batches} batch(es): "
          "divide by that for one batch; SQ cycle counters are quad-cycles)\n")
with open(os.path.join(out, f"{prefix}_pmc_all_kernels.txt"), "w") as f:
    f.write(header)
    for (k, c), v in sorted(rows.items()):
        f.write(f"{k[:32]:34s} {c:24s} {v:.4e}\n")
with open(os.path.join(out, f"{prefix}_pmc_extend.txt"), "w") as f:   # (r6: the lockstep fragment kernel and what it declines)
    f.write(header)
    for (k, c), v in sorted(rows.items()):
        if k == "k_extend" or k.startswith("k_extend_slab") or k.startswith("k_build_anchors"):
            f.write(f"{k[:32]:34s} {c:24s} {v:.4e}\n")
if os.path.exists(os.path.join(src, "frag_stamps.txt")):
    shutil.copy(os.path.join(src, "frag_stamps.txt"), os.path.join(out, f"{prefix}_stamps_extend.txt"))
with open(os.path.join(out, f"{prefix}_pmc_long_extend.txt"), "w") as f:
    f.write(header)
    for (k, c), v in sorted(rows.items()):
        if k.startswith("k_long_extend<1"):
            f.write(f"{k[:32]:34s} {c:24s} {v:.4e}\n")

# HBM traffic per kernel and step
traffic = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), bench.py --no-cpu-baseline --steps 1 --warmup 0, cfg2",
           "note": "KB as rocprofv3 reports them (TCC_EA request counters x request size); per the MI355X guide gfx950 FETCH_SIZE under-reports wide coalesced streaming reads by 2x and is uncalibrated for other widths - these kernels issue 4-8 B scattered accesses, so the figures are used as reported",
           "batches_in_pass": batches,
           "kernels": {}}
for sub, counter in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    agg, launches = sums(os.path.join(src, sub, "p_counter_collection.csv"))
    for (k, c), v in agg.items():
        if c != counter:
            continue
        entry = traffic["kernels"].setdefault(k, {})
        entry[f"{counter}_KB_per_step"] = v
        entry["launches_per_step"] = launches[k]
json.dump(traffic, open(os.path.join(out, f"{prefix}_pmc_traffic.json"), "w"), indent=1)

# section stamps of the dominant kernel
stamps = [l for l in open(os.path.join(src, "stamps.txt")) if l.startswith("[gc stamps]")]
if stamps:
    open(os.path.join(out, f"{prefix}_stamps_long_extend.txt"), "w").write("".join(stamps[-11:]))   # the timed step (a warm-up step prints too)
print("wrote", sorted(f for f in os.listdir(out) if f.startswith(prefix)))
