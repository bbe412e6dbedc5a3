#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counter_collection.csv values per kernel: python scripts/pmc_summary.py <csv> [kernel substring]"""
import csv, collections, sys
agg = collections.defaultdict(float)
for r in csv.DictReader(open(sys.argv[1])):
    agg[(r['Kernel_Name'].replace('gcdev::', '').split('(')[0][:32], r['Counter_Name'])] += float(r['Counter_Value'])
want = sys.argv[2] if len(sys.argv) > 2 else ''
for (k, c), v in sorted(agg.items()):
    if want in k:
        print(f"{k:34s} {c:24s} {v:.4e}")
