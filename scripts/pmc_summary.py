#!/usr/bin/env python3
"""Condenses the rocprofv3 outputs of scripts/profile.sh into two small CSVs:
gpurun_out/TAG_kernel_stats.csv (per-kernel calls / total / average duration from --kernel-trace)
and gpurun_out/TAG_pmc.csv (per-kernel mean of every collected counter)."""
import csv, glob, os, sys, collections

tag, cmd = sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else ""
out = "gpurun_out"

def short(name):
    """Kernel name without arguments and template lists; the two builds of the fused PG kernel
    (select-free / exact stop test, of which one returns at once) stay apart."""
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    head = name.split("(")[0].strip()
    base = head.split("<")[0]
    if "lane_pg_fused_kernel" in base or "ub_pg_kernel" in base or "ubg_pg_kernel" in base:
        args = head[len(base):]
        last = args.rstrip(">").rstrip().split(",")[-1].strip()   # bool FAST, or ub_pg_kernel's int MODE (0 exact, 1 mask, 2 moved)
        base += "<fast>" if last in ("true", "2") else ("<mask>" if last == "1" else "<exact>")
    if base.endswith("group_pg_kernel") and head[len(base):].lstrip("<").startswith("float"):
        # fp32 GROUP: two builds of the stop test (MOVED = the last argument), launched back to back; one returns at once
        last = head[len(base):].rstrip(">").rstrip().split(",")[-1].strip()
        base += "<f32 moved>" if last == "true" else "<f32 mask>"
    return base

# kernel trace
rows = collections.defaultdict(list)
for f in glob.glob(f"{out}/prof_{tag}_trace/*/*_kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        rows[short(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
with open(f"{out}/{tag}_kernel_stats.csv", "w") as fh:
    fh.write(f"# rocprofv3 --kernel-trace --stats: {cmd}\n")
    fh.write("kernel,calls,total_ms,avg_ms,min_ms,max_ms\n")
    for k, v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
        fh.write(f"{k},{len(v)},{sum(v)/1e6:.4f},{sum(v)/len(v)/1e6:.4f},{min(v)/1e6:.4f},{max(v)/1e6:.4f}\n")

# counters
vals = collections.defaultdict(list)
for name in ("fetch", "write", "sq", "sq2"):
    for f in glob.glob(f"{out}/prof_{tag}_{name}/*/*_counter_collection.csv"):
        per = collections.defaultdict(float)
        for r in csv.DictReader(open(f)):
            per[(short(r["Kernel_Name"]), r["Counter_Name"], r["Dispatch_Id"])] += float(r["Counter_Value"])
        for (k, c, _), v in per.items():
            vals[(k, c)].append(v)
with open(f"{out}/{tag}_pmc.csv", "w") as fh:
    fh.write(f"# rocprofv3 PMC passes (each counter set in its own run, with --kernel-trace): {cmd}\n")
    fh.write("# FETCH_SIZE / WRITE_SIZE in the counter's unit (KiB); SQ_* summed over the chip (SQ cycle counters tick every 4 shader cycles); mean over dispatches\n")
    fh.write("kernel,counter,dispatches,mean_value\n")
    for (k, c), v in sorted(vals.items()):
        if k.startswith("tpc::") or "tpc" in k:
            fh.write(f"{k},{c},{len(v)},{sum(v)/len(v):.6g}\n")
print(open(f"{out}/{tag}_kernel_stats.csv").read())
print(open(f"{out}/{tag}_pmc.csv").read())
