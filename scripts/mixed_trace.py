#!/usr/bin/env python3
"""Diagnostic: kernel timeline of one mixed-horizon batch (BASELINE config 5) from a rocprofv3 --kernel-trace CSV.
    rocprofv3 --kernel-trace --output-format csv -d OUT -- python3 scripts/mixed_horizons.py f64
    python scripts/mixed_trace.py OUT        (prints start / end of every kernel of the LAST mixed call, in ms)"""
import csv, glob, os, sys
path = [p for p in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)][0]
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last call starts at the last mixed_bin_kernel
idx = max(i for i, r in enumerate(rows) if "mixed_bin_kernel" in r["Kernel_Name"])
t0 = int(rows[idx]["Start_Timestamp"])
for r in rows[idx:]:
    name = r["Kernel_Name"]
    short = name.split("(")[0].replace("void tpc::", "").replace("tpc::", "")[:70]
    print(f"{(int(r['Start_Timestamp']) - t0) / 1e6:8.3f} .. {(int(r['End_Timestamp']) - t0) / 1e6:8.3f} ms  {short}  grid {r.get('Grid_Size', r.get('Grid_Size_X', '?'))}")
