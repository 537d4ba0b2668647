#!/usr/bin/env python3
"""Per-dispatch kernel durations from a rocprofv3 --kernel-trace database (…_results.db), in launch order."""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kt = [t for t in tabs if 'kernel_dispatch' in t][0]
ks = [t for t in tabs if 'kernel_symbol' in t][0]
q = f"select s.kernel_name, d.start, d.end, d.grid_size_x from {kt} d join {ks} s on d.kernel_id=s.id order by d.start"
pat = re.compile(sys.argv[2]) if len(sys.argv) > 2 else None
for name, st, en, g in cur.execute(q):
    if pat and not pat.search(name):
        continue
    print(f"{(en - st) / 1e3:10.1f} us  grid {g:9d}  {name[:90]}")
