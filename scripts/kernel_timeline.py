#!/usr/bin/env python3
"""Print the per-launch kernel timeline of a rocprofv3 --kernel-trace run (results .db)."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
c = db.cursor()
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
ix = {n: i for i, n in enumerate(cols)}
t0 = None
last_end = None
for r in c.execute("select * from kernels order by start"):
    n = r[ix['name']]
    s, e = r[ix['start']], r[ix['end']]
    if t0 is None:
        t0 = s
    short = n.replace('gcdev::', '').split('(')[0][:28]
    print(f"{short:28s} start {(s - t0) / 1e6:9.2f} ms  dur {(e - s) / 1e6:9.3f} ms  grid {r[ix['grid_x']]:8d}  lds {r[ix['lds_size']]:6d}  vgpr {r[ix['vgpr_count']]}")
