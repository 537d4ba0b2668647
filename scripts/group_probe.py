#!/usr/bin/env python3
"""Diagnostic: the GROUP family (G lanes per instance) -- parity against the oracle and kernel times beside WAVE / LANE_FMA.
    python scripts/group_probe.py [f64|f32] H G[,G...] n[,n...] [check]    (TPC_MPC_LIB picks the build)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from trajectory_controller_amd import MpcSolver, capi
from trajectory_controller_amd.synth import compact_inputs
args = sys.argv[1:]
dtype = args.pop(0) if args and args[0] in ("f64", "f32") else "f64"
H = int(args[0]); Gs = [int(x) for x in args[1].split(",")]; ns = [int(x) for x in args[2].split(",")]
check = len(args) > 3 and args[3] == "check"
waves = [0]
tdt = torch.float64 if dtype == "f64" else torch.float32
for n in ns:
    v, dy, dphi = compact_inputs(H, n)
    tv, ty, tp = (torch.from_numpy(a).to("cuda", dtype=tdt) for a in (v, dy, dphi))
    ref = None
    if check:
        from oracle.bindings import Oracle
        ref = Oracle("f64" if dtype == "f64" else "f32").solve_compact(H, v.astype(np.float64 if dtype == "f64" else np.float32), dy.astype(v.dtype if dtype == "f64" else np.float32), dphi.astype(v.dtype if dtype == "f64" else np.float32), nthreads=16)
    rows = []
    for algo, G, W in [("wave", 0, 0), ("lane_fma", 0, 0)] + [("group", G, W) for G in Gs for W in waves]:
        if algo == "wave" and n > 32768 and H >= 20:
            continue
        with MpcSolver(horizon=H, algo=algo, dtype=dtype) as s:
            if G:
                s.set_option(capi.OPT_GROUP_LANES, G)
            s.set_profiling(True)
            best = (1e9, 0, 0)
            for _ in range(3):
                f, r, it = s.solve_batch_compact(tv, ty, tp, want_iters=True, want_flags=False)
                k1, k2, ran = s.last_kernel_times()
                if k1 + k2 < best[0]: best = (k1 + k2, k1, k2)
            msg = f"{dtype} H={H} n={n:6d} {algo:8s} G={G} W={W} ran={ran}: {best[0]:8.3f} ms (cd {best[1]:.3f} pg {best[2]:.3f})"
            if algo != "wave":
                wi, rb = s.last_lane_stats()
                msg += f" wave-iters {wi} refills {rb}"
            if ref is not None:
                f, r, it = f.cpu().numpy(), r.cpu().numpy(), it.cpu().numpy()
                err = max(np.abs(f - ref[0]).max(), np.abs(r - ref[1]).max())
                msg += f"  max|du| {err:.2e} iters equal {np.mean(it == ref[2]):.6f}"
            print(msg, flush=True)
