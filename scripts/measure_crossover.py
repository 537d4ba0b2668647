#!/usr/bin/env python3
"""Re-measure AUTO's crossovers (compact form) and regenerate trajectory_controller_amd/csrc/auto_table.h.

    python scripts/measure_crossover.py [--write] [--general] [--out profiles/rNN_crossover.txt] [f64|f32 ...] [H ...]

For every dtype and horizon with GROUP kernels it times WAVE, GROUP with 8 / 4 / 2 lanes per instance and LANE_FMA
(kernel time through the library's own HIP events, best of three) on the BASELINE input distribution at a ladder of
batch sizes, prints the table, and derives for each family the batch size below which it is the fastest: families are
ordered WAVE -> GROUP 8 -> 4 -> 2 -> LANE_FMA as the batch grows, and a crossover is the geometric mean of the last size
the smaller family wins and the first size the next one does.  With --write the header is rewritten (rebuild the library
afterwards: make -C trajectory_controller_amd/csrc).  --general measures the general form instead (two inputs, fp64, cold
starts; the one-lane family is LANE_FMA up to N = 20 and LANE beyond) and writes the rows with form = 1."""
import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from trajectory_controller_amd import MpcSolver, capi
from trajectory_controller_amd.synth import compact_inputs, general_inputs

LADDER = [1024, 1536, 2048, 3072, 4096, 6144, 8192, 12288, 16384, 24576, 32768, 49152, 65536, 98304, 131072, 196608, 262144]
BUILT = {10: (4, 2), 20: (8, 4, 2), 30: (8, 4, 2), 40: (8, 4, 2)}
args = sys.argv[1:]
write = "--write" in args
general = "--general" in args
out_path = None
if "--out" in args:
    out_path = args[args.index("--out") + 1]
    del args[args.index("--out"):args.index("--out") + 2]
args = [a for a in args if a not in ("--write", "--general")]
if general:
    BUILT = {10: (4, 2), 20: (8, 4, 2), 30: (8, 4), 40: (8, 4)}
    LADDER = [n for n in LADDER if n <= 131072]
    GN = ["A", "B", "C", "Q", "R", "lo", "hi", "x0", "targets"]
dtypes = [a for a in args if a in ("f64", "f32")] or (["f64"] if general else ["f64", "f32"])
hs = [int(a) for a in args if a.isdigit()] or [10, 20, 30, 40]
lines = []
def say(msg):
    print(msg, flush=True)
    lines.append(msg)

def time_one(H, dtype, algo, G, tv, ty, tp):
    with MpcSolver(horizon=H, algo=algo, dtype=dtype) as s:
        if G:
            s.set_option(capi.OPT_GROUP_LANES, G)
        s.set_profiling(True)
        best = 1e9
        for _ in range(3):
            if general:
                s.solve_batch_general(*tv, inputs=2)
            else:
                s.solve_batch_compact(tv, ty, tp, want_flags=False)
            k1, k2, ran = s.last_kernel_times()
            best = min(best, k1 + k2)
    return best

rows = {}
for dtype in dtypes:
    tdt = torch.float64 if dtype == "f64" else torch.float32
    for H in hs:
        fams = ["wave"] + [f"g{g}" for g in BUILT[H]] + ["lane_fma"]
        table = {}
        for n in LADDER:
            if general:
                gi = general_inputs(H, n, I=2)
                tv = [torch.from_numpy(np.ascontiguousarray(gi[k].reshape(n, -1).T)).to("cuda", dtype=tdt) for k in GN]
                ty = tp = None
            else:
                v, dy, dphi = compact_inputs(H, n)
                tv, ty, tp = (torch.from_numpy(a).to("cuda", dtype=tdt) for a in (v, dy, dphi))
            t = {}
            if n <= 32768:
                t["wave"] = time_one(H, dtype, "wave", 0, tv, ty, tp)
            for g in BUILT[H]:
                t[f"g{g}"] = time_one(H, dtype, "group", g, tv, ty, tp)
            t["lane_fma"] = time_one(H, dtype, "lane_fma", 0, tv, ty, tp)
            table[n] = t
            best = min(t, key=t.get)
            say(f"{'general ' if general else ''}{dtype} H={H:2d} n={n:6d}: " + "  ".join(f"{f} {t[f]:7.3f}" if f in t else f"{f}    --  " for f in fams) + f"  -> {best}")
        # crossovers in family order; a family that never wins gets its predecessor's bound
        winners = [min(table[n], key=table[n].get) for n in LADDER]
        bounds, prev = {}, 0
        for i, f in enumerate(fams[:-1]):
            later = fams[i + 1:]
            # first ladder index from which some later family wins for good
            idx = next((j for j in range(len(LADDER)) if all(w in later for w in winners[j:])), len(LADDER))
            if idx == 0:
                b = prev
            elif idx == len(LADDER):
                b = LADDER[-1] * 2
            else:
                b = int(round(math.sqrt(LADDER[idx - 1] * LADDER[idx])))
            b = max(b, prev)
            bounds[f] = prev = b
        rows[(dtype, H)] = (bounds.get("wave", 0), bounds.get("g8", bounds.get("wave", 0)), bounds.get("g4", 0), bounds.get("g2", bounds.get("g4", 0)))
        say(f"   => {dtype} H={H}: WAVE below {rows[(dtype, H)][0]}, GROUP 8 below {rows[(dtype, H)][1]}, 4 below {rows[(dtype, H)][2]}, 2 below {rows[(dtype, H)][3]}, LANE_FMA from there")

if out_path:
    with open(os.path.join(ROOT, out_path), "w") as f:
        f.write("\n".join(lines) + "\n")
if write:
    hdr = os.path.join(ROOT, "trajectory_controller_amd", "csrc", "auto_table.h")
    src = open(hdr).read()
    head, tail = src[:src.index("constexpr AutoRow kAutoTable[] = {")], src[src.index("};\n// clang-format on"):]
    old = {}
    for line in src[len(head):len(src) - len(tail)].splitlines():
        line = line.strip()
        if line.startswith("{"):
            nums = [int(x) for x in line.strip("{},").split(",")]
            old[(nums[0], ("f64", "f32")[nums[1]], nums[2])] = tuple(nums[3:])
    old.update({(1 if general else 0, d, H): r for (d, H), r in rows.items()})
    body = "constexpr AutoRow kAutoTable[] = {\n" + "".join(
        f"    {{{form}, {0 if d == 'f64' else 1}, {H}, {r[0]:6d}, {r[1]:6d}, {r[2]:6d}, {r[3]:6d}}},\n" for (form, d, H), r in sorted(old.items()))
    open(hdr, "w").write(head + body + tail)
    print("rewrote", hdr)
