#!/usr/bin/env python3
"""Diagnostic: LANE / LANE_FMA kernel times for chosen horizons: scripts/lane_h.py DTYPE H[,H...] [n] [lane|lane_fma]   (TPC_MPC_LIB picks the build)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from trajectory_controller_amd import MpcSolver
from trajectory_controller_amd.synth import compact_inputs
dtype = sys.argv[1] if len(sys.argv) > 1 else "f64"
hs = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [20]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 262144
algo = sys.argv[4] if len(sys.argv) > 4 else "lane"
tdt = torch.float64 if dtype == "f64" else torch.float32
tag = os.environ.get("TPC_MPC_LIB", "default").split("/")[-2] if "TPC_MPC_LIB" in os.environ else "default"
for H in hs:
    v, dy, dphi = (torch.from_numpy(a).to("cuda", dtype=tdt) for a in compact_inputs(H, n))
    with MpcSolver(horizon=H, dtype=dtype, algo=algo) as s:
        s.set_profiling(True)
        best = None
        for _ in range(3):
            f, r, it = s.solve_batch_compact(v, dy, dphi, want_iters=True, want_flags=False)
            k1, k2, _ = s.last_kernel_times()
            best = (k1, k2) if best is None or k1 + k2 < sum(best) else best
        wi, rb = s.last_lane_stats()
        k1, k2 = best
        print(f"{tag:10s} {algo} {dtype} H={H:2d} n={n}: cd {k1:7.3f} ms  pg {k2:8.3f} ms  {n/(k1+k2)/1e3:8.2f} M solves/s  "
              f"wave-iters {wi}  chk {float(f.double().sum()):.12f} {float(it.double().sum()):.0f}", flush=True)
