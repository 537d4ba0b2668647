#!/usr/bin/env python3
"""Re-measure AUTO's crossovers (compact form) and regenerate trajectory_controller_amd/csrc/auto_table.h.

    python scripts/measure_crossover.py [--write] [--general] [--out profiles/rNN_crossover.txt] [f64|f32 ...] [H ...]
    python scripts/measure_crossover.py --from-record profiles/rNN_crossover.txt[,more.txt] [--write]   (no GPU: re-derive the rows)
    python scripts/measure_crossover.py --ladder 262144,393216,524288 --out profiles/rNN_crossover_top.txt f64 40   (extend a record)

For every dtype and horizon with GROUP kernels it times WAVE, GROUP with 8 / 4 / 2 lanes per instance and LANE_FMA
(kernel time through the library's own HIP events, best of three) on the BASELINE input distribution at a ladder of
batch sizes, prints the table, and derives for each family the batch size below which it is the fastest: families are
ordered WAVE -> GROUP 8 -> 4 -> 2 -> LANE_FMA as the batch grows, and a crossover is the geometric mean of the last size
the smaller family wins and the first size the next one does.  With --write the header is rewritten (rebuild the library
afterwards: make -C trajectory_controller_amd/csrc).  --general measures the general form instead (two inputs, fp64, cold
starts; the one-lane family is LANE_FMA up to N = 20 and LANE beyond) and writes the rows with form = 1."""
import math, os, re, sys
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
if "--ladder" in args:   # batch sizes to measure instead of the default ladder (e.g. the top end only, to extend a record)
    LADDER = [int(x) for x in args[args.index("--ladder") + 1].split(",")]
    del args[args.index("--ladder"):args.index("--ladder") + 2]
    CUSTOM_LADDER = True
else:
    CUSTOM_LADDER = False
record = None
if "--from-record" in args:
    record = args[args.index("--from-record") + 1]
    del args[args.index("--from-record"):args.index("--from-record") + 2]
if "--out" in args:
    out_path = args[args.index("--out") + 1]
    del args[args.index("--out"):args.index("--out") + 2]
args = [a for a in args if a not in ("--write", "--general")]
if general:
    BUILT = {10: (4, 2), 20: (8, 4, 2), 30: (8, 4), 40: (8, 4)}
    if not CUSTOM_LADDER:
        LADDER = [n for n in LADDER if n <= 131072]
    GN = ["A", "B", "C", "Q", "R", "lo", "hi", "x0", "targets"]
dtypes = [a for a in args if a in ("f64", "f32")] or (["f64"] if general else ["f64", "f32"])
hs = [int(a) for a in args if a.isdigit()] or [10, 20, 30, 40]
lines = []
def say(msg):
    print(msg, flush=True)
    lines.append(msg)

def time_one(H, dtype, algo, G, tv, ty, tp, per_simd=0):
    with MpcSolver(horizon=H, algo=algo, dtype=dtype) as s:
        if G:
            s.set_option(capi.OPT_GROUP_LANES, G)
        if per_simd:   # the persistent grid pinned to one / two wavefronts per SIMD (kernels built for two: fp32, compact form)
            s._check(s._lib.tpc_mpc_x_set_group_share(s._h, per_simd * 4 * torch.cuda.get_device_properties(0).multi_processor_count, 0))
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

def PAIRED(dtype):
    """GROUP kernels built for two wavefronts per SIMD: fp32, compact form (csrc/mpc_group.h, GroupPlan::occ)."""
    return dtype == "f32" and not general


NEVER = 1 << 40   # "this family is never overtaken": beyond any batch that fits a GPU


def derive(table, fams):
    """Crossovers in family order from {n: {family: ms}}; a family that never wins gets its predecessor's bound.
    A family still ahead at the top of the ladder is extrapolated with the per-instance cost of the last two sizes
    (both sides are throughput-bound there, so time is linear in n): overtaken where the lines cross, or NEVER."""
    ladder = sorted(table)
    pair_from = NEVER
    if any(k.endswith("p") for t in table.values() for k in t):
        # columns "gNp": the same kernel with the grid at two wavefronts per SIMD.  pair_from = where the best paired
        # GROUP time is ahead of the best lone one for good; the table the crossovers are derived from then holds, per
        # size, the times of the rule in force there.
        best = lambda t, sfx: min((v for k, v in t.items() if k.startswith("g") and k.endswith("p") == (sfx == "p")), default=1e30)
        ahead = [best(table[n], "p") < best(table[n], "") for n in ladder]
        idx = next((j for j in range(len(ladder)) if all(ahead[j:])), len(ladder))
        pair_from = 0 if idx == 0 else (NEVER if idx == len(ladder) else int(round(math.sqrt(ladder[idx - 1] * ladder[idx]))))
        eff = {}
        for n in ladder:
            eff[n] = {k: v for k, v in table[n].items() if not k.startswith("g")}
            for k, v in table[n].items():
                if k.startswith("g") and k.endswith("p") == (n >= pair_from):
                    eff[n][k.rstrip("p")] = v
        table = eff
    winners = [min(table[n], key=table[n].get) for n in ladder]
    bounds, prev = {}, 0
    for i, f in enumerate(fams[:-1]):
        later = fams[i + 1:]
        # first ladder index from which some later family wins for good
        idx = next((j for j in range(len(ladder)) if all(w in later for w in winners[j:])), len(ladder))
        if idx == 0:
            b = prev
        elif idx == len(ladder) and len(ladder) < 2:
            b = NEVER   # (a single size says nothing about cost per further instance: leave the leader in place)
        elif idx == len(ladder):
            n1, n0 = ladder[-1], ladder[-2]
            w = winners[-1]
            slope = lambda fam: (table[n1][fam] - table[n0][fam]) / (n1 - n0)
            b = NEVER
            for g in later:
                if g == w or g not in table[n1] or g not in table[n0]:
                    continue
                if slope(w) > slope(g):
                    b = min(b, int(n1 + (table[n1][g] - table[n1][w]) / (slope(w) - slope(g))))
            b = max(b, n1)
        else:
            b = int(round(math.sqrt(ladder[idx - 1] * ladder[idx])))
        b = max(b, prev)
        bounds[f] = prev = b
    return (bounds.get("wave", 0), bounds.get("g8", bounds.get("wave", 0)), bounds.get("g4", 0), bounds.get("g2", bounds.get("g4", 0)), pair_from)


def summary(dtype, H, r):
    word = lambda b: "never overtaken" if b >= NEVER else f"below {b}"
    pair = "" if r[4] >= NEVER else f"; GROUP grid two wavefronts per SIMD from {r[4]}"
    return f"   => {dtype} H={H}: WAVE {word(r[0])}, GROUP 8 {word(r[1])}, 4 {word(r[2])}, 2 {word(r[3])}, LANE_FMA from there{pair}"


def from_record(path):
    """Re-derive the rows from a committed record of this script (no GPU): profiles/rNN_crossover*.txt."""
    tables, is_general = {}, False
    for line in open(path):
        mt = re.match(r"(general )?(f64|f32) H=\s*(\d+) n=\s*(\d+): (.*?)\s+-> ", line)
        if not mt:
            continue
        is_general = is_general or bool(mt.group(1))
        t = {f: float(x) for f, x in re.findall(r"(wave|g8p|g4p|g2p|g8|g4|g2|lane_fma)\s+([0-9.]+)", mt.group(5))}
        tables.setdefault((mt.group(2), int(mt.group(3))), {})[int(mt.group(4))] = t
    return tables, is_general


rows = {}
if record:
    tables, general = {}, False
    for rec in record.split(","):   # several records: a later one adds sizes to (or replaces sizes of) an earlier one
        more, g = from_record(rec if os.path.isabs(rec) else os.path.join(ROOT, rec))
        general = general or g
        for key, tab in more.items():
            tables.setdefault(key, {}).update(tab)
    if general:
        BUILT = {10: (4, 2), 20: (8, 4, 2), 30: (8, 4), 40: (8, 4)}
    for (dtype, H), table in sorted(tables.items()):
        rows[(dtype, H)] = derive(table, ["wave"] + [f"g{g}" for g in BUILT[H]] + ["lane_fma"])
        say(summary(dtype, H, rows[(dtype, H)]))
    dtypes = []
for dtype in dtypes:
    tdt = torch.float64 if dtype == "f64" else torch.float32
    for H in hs:
        fams = ["wave"] + [f"g{g}" for g in BUILT[H]] + ["lane_fma"]
        cols = ["wave"] + [c for g in BUILT[H] for c in ([f"g{g}", f"g{g}p"] if PAIRED(dtype) else [f"g{g}"])] + ["lane_fma"]
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
                if PAIRED(dtype):
                    t[f"g{g}"] = time_one(H, dtype, "group", g, tv, ty, tp, per_simd=1)
                    t[f"g{g}p"] = time_one(H, dtype, "group", g, tv, ty, tp, per_simd=2)
                else:
                    t[f"g{g}"] = time_one(H, dtype, "group", g, tv, ty, tp)
            t["lane_fma"] = time_one(H, dtype, "lane_fma", 0, tv, ty, tp)
            table[n] = t
            best = min(t, key=t.get)
            say(f"{'general ' if general else ''}{dtype} H={H:2d} n={n:6d}: " + "  ".join(f"{f} {t[f]:7.3f}" if f in t else f"{f}    --  " for f in cols) + f"  -> {best}")
        rows[(dtype, H)] = derive(table, fams)
        say(summary(dtype, H, rows[(dtype, H)]))

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
            old[(nums[0], ("f64", "f32")[nums[1]], nums[2])] = tuple(nums[3:]) if len(nums) >= 8 else tuple(nums[3:]) + (NEVER,)
    old.update({(1 if general else 0, d, H): r for (d, H), r in rows.items()})
    body = "constexpr AutoRow kAutoTable[] = {\n" + "".join(
        f"    {{{form}, {0 if d == 'f64' else 1}, {H}, {r[0]:6d}, {r[1]:6d}, {r[2]:6d}, {r[3]:6d}, {r[4]:6d}}},\n" for (form, d, H), r in sorted(old.items()))
    open(hdr, "w").write(head + body + tail)
    print("rewrote", hdr)
