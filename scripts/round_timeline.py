#!/usr/bin/env python3
"""Critical path of the whole-read pass from a rocprofv3 --kernel-trace CSV: for every queue that runs k_long_extend, the kernels of one
pass in launch order with their start, duration and the idle gap before them; and per pass the sum of extension time, helper time and gaps.
Usage: round_timeline.py kernel_trace.csv [pass index to print in full]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
show = int(sys.argv[2]) if len(sys.argv) > 2 else 2
name = lambda r: r["Kernel_Name"].replace("gcdev::", "").replace("void ", "").split("(")[0][:34]
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
t0 = min(r["s"] for r in rows)
byq = collections.defaultdict(list)
for r in rows:
    byq[r["Queue_Id"]].append(r)
for q, rs in byq.items():
    rs.sort(key=lambda r: r["s"])
    if not any("k_long_extend" in r["Kernel_Name"] for r in rs):
        continue
    # passes: split at k_long_init
    passes, cur = [], []
    for r in rs:
        if "k_long_init" in r["Kernel_Name"] and cur:
            passes.append(cur); cur = []
        cur.append(r)
    passes.append(cur)
    print(f"queue {q}: {len(passes)} passes")
    for pi, p in enumerate(passes):
        ext = sum(r["e"] - r["s"] for r in p if "k_long_extend" in r["Kernel_Name"]) / 1e6
        helpers = sum(r["e"] - r["s"] for r in p if "k_long_extend" not in r["Kernel_Name"]) / 1e6
        span = (p[-1]["e"] - p[0]["s"]) / 1e6
        gaps = span - ext - helpers
        print(f"  pass {pi}: span {span:8.2f} ms  extension kernels {ext:8.2f}  other kernels {helpers:8.2f}  idle between kernels {gaps:8.2f}  ({len(p)} launches)")
        if pi == show:
            last = None
            for r in p:
                gap = (r["s"] - last) / 1e6 if last is not None else 0.0
                print(f"      {name(r):34s} start {(r['s'] - t0) / 1e6:10.2f}  dur {(r['e'] - r['s']) / 1e6:8.3f}  gap before {gap:8.3f}")
                last = r["e"]
