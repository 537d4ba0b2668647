#!/usr/bin/env python3
"""profiles/traffic.json from the PMC summaries scripts/profile.sh left in profiles/ (or gpurun_out/):
HBM bytes per launch of each profiled kernel = (k * FETCH_SIZE + WRITE_SIZE) * 1024, where k is the
gfx950 read correction CALIBRATED PER ACCESS PATTERN on a known byte count, as
MI355X_MICROARCH.md (HBM) asks:
  * wide coalesced streaming reads (lane_cd_kernel's inputs: 3 x n x 8 B known, FETCH_SIZE reports
    half of it)                                                              -> k = 2
  * one 8-byte load per lane at a 336-byte stride (lane_pg_fused_kernel re-reading the CD kernel's
    per-instance records: records + inputs + queue = 364 B x n known, FETCH_SIZE reports 1.05 x that) -> k = 1
Kernels with no calibration of their own get k = 1 and say so."""
import csv, json, os, sys
src = sys.argv[1] if len(sys.argv) > 1 else "profiles"
rnd = sys.argv[2] if len(sys.argv) > 2 else "r05"
CONFIGS = {   # tag -> (kernel in the csv, bench key, k, note)
    # the hand-written kernel reads each instance's 400-byte record with 25 dwordx4 loads + one 4-byte queue entry: 105.9 MB
    # known, FETCH_SIZE reports 0.64 of it -> k = 1.57 (between the wide-streaming k = 2 and the strided 8-byte k = 1)
    "headline": ("tpc::ub_pg_asm_kernel", "ub_pg_asm_kernel_f64_H20_n262144", 1.57, "25 x 16 B record loads per instance: k calibrated on this kernel's own known read set (404 B x n)"),
    "bitexact": ("tpc::lane_pg_fused_kernel<fast>", "lane_pg_fused_kernel_f64_H20_n262144", 1, "strided record reads: k=1 (calibrated)"),
    # (fp32 at 262 144 x N=20 under AUTO is GROUP since the end of round 4: two lanes per instance, two wavefronts per SIMD)
    "fp32": ("tpc::group_pg_kernel<f32 moved>", "group_pg_kernel_f32_H20_n262144", 1, "chunked record reads: k=1 (uncalibrated)"),
    "group_f32_h20_256k": ("tpc::group_pg_kernel<f32 moved>", "group_pg_kernel_f32_H20_n262144_lane_h", 1, "the same kernel through scripts/lane_h.py: k=1 (uncalibrated)"),
    "group_f32_h40_256k": ("tpc::group_pg_kernel<f32 mask>", "group_pg_kernel_f32_H40_n262144", 1, "chunked record reads: k=1 (uncalibrated)"),
    "config2": ("tpc::wave_pair_queue_kernel", "wave_kernel_f64_H10_n4096", 1, "broadcast loads of 3 scalars per wavefront + the queue order: uncalibrated, k=1"),
    "general": ("tpc::lane_pg_fused_kernel<fast>", "lane_pg_fused_kernel_general_I2_f64_H20_n262144", 1, "strided record reads + SoA model loads: k=1"),
    "group_h20": ("tpc::group_pg_kernel", "group_pg_kernel_f64_H20_n16384", 1, "chunked record reads, 8 lanes per record: k=1 (uncalibrated)"),
    "group_h20_64k": ("tpc::group_pg_kernel", "group_pg_kernel_f64_H20_n65536", 1, "chunked record reads: k=1 (uncalibrated)"),
    "group_h40": ("tpc::group_pg_kernel", "group_pg_kernel_f64_H40_n16384", 1, "chunked record reads: k=1 (uncalibrated)"),
    "group_h10": ("tpc::group_pg_kernel", "group_pg_kernel_f64_H10_n32768", 1, "chunked record reads: k=1 (uncalibrated)"),
    "group_h40_256k": ("tpc::group_pg_kernel", "group_pg_kernel_f64_H40_n262144", 1, "chunked record reads: k=1 (uncalibrated)"),
    "group_h30_256k": ("tpc::group_pg_kernel", "group_pg_kernel_f64_H30_n262144", 1, "chunked record reads: k=1 (uncalibrated)"),
    "groupg_h40": ("tpc::groupg_pg_kernel", "groupg_pg_kernel_general_I2_f64_H40_n16384", 1, "chunked record reads + SoA model loads: k=1 (uncalibrated)"),
    "lanex_h40": ("tpc::lanex_pg_kernel", "lanex_pg_kernel_f64_H40_n16384", 1, "chunked record reads: k=1 (uncalibrated)"),
    "lanex_h20": ("tpc::lanex_pg_kernel", "lanex_pg_kernel_f64_H20_n16384", 1, "chunked record reads: k=1 (uncalibrated)"),
    "groupg_h20": ("tpc::groupg_pg_kernel", "groupg_pg_kernel_general_I2_f64_H20_n16384", 1, "chunked record reads + SoA model loads: k=1 (uncalibrated)"),
    "generalfma": ("tpc::ubg_pg_kernel<fast>", "ubg_pg_kernel_general_I2_f64_H20_n262144", 1, "strided record reads + SoA model loads: k=1"),
}
import hashlib
LIB = os.path.join("trajectory_controller_amd", "lib", "libtpc_mpc.so")
out = {"_how": __doc__.strip(), "_raw_KiB": {},
       # bench.py reports a figure from this file only when it runs THIS binary
       "_library_sha256": hashlib.sha256(open(LIB, "rb").read()).hexdigest() if os.path.exists(LIB) else None}
for tag, (kern, key, k, note) in CONFIGS.items():
    f = os.path.join(src, f"{rnd}_{tag}_pmc.csv")
    if not os.path.exists(f):
        continue
    vals = {}
    for row in csv.reader(l for l in open(f) if not l.startswith("#")):
        if len(row) == 4 and row[0] == kern:
            vals[row[1]] = float(row[3])
    if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
        out[key] = int((k * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024)
        out["_raw_KiB"][key] = {"FETCH_SIZE": vals["FETCH_SIZE"], "WRITE_SIZE": vals["WRITE_SIZE"], "read_correction": k, "note": note}
    # the CD kernel of the bit-exact run documents the k = 2 calibration
    if tag == "bitexact":
        cd = {}
        for row in csv.reader(l for l in open(f) if not l.startswith("#")):
            if len(row) == 4 and row[0] == "tpc::lane_cd_kernel":
                cd[row[1]] = float(row[3])
        if "FETCH_SIZE" in cd:
            out["_raw_KiB"]["lane_cd_kernel_f64_H20_n262144"] = {"FETCH_SIZE": cd["FETCH_SIZE"], "WRITE_SIZE": cd.get("WRITE_SIZE"),
                                                                 "known_input_bytes": 3 * 262144 * 8, "read_correction": 2}
            out["lane_cd_kernel_f64_H20_n262144"] = int((2 * cd["FETCH_SIZE"] + cd.get("WRITE_SIZE", 0)) * 1024)
json.dump(out, open(os.path.join("profiles", "traffic.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if not k.startswith("_")}, indent=1))
