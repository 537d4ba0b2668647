#!/usr/bin/env python3
"""Diagnostic: cost of ONE solver iteration of the LANE projected-gradient kernels.
All instances are identical, one per lane of a full chip (65536), so every lane runs the same number
of iterations and no refill happens mid-run: kernel time / iterations = time per wave-iteration."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from trajectory_controller_amd import MpcSolver

H = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dtype = sys.argv[2] if len(sys.argv) > 2 else "f64"
n = int(sys.argv[3]) if len(sys.argv) > 3 else 65536
tdt = torch.float64 if dtype == "f64" else torch.float32
v = torch.full((n,), 3.9, dtype=tdt, device="cuda")
dy = torch.full((n,), 0.45, dtype=tdt, device="cuda")
dphi = torch.full((n,), -0.55, dtype=tdt, device="cuda")
s = MpcSolver(horizon=H, dtype=dtype, algo=(sys.argv[4] if len(sys.argv) > 4 else "lane"))
s.set_profiling(True)
for rep in range(3):
    f, r, it = s.solve_batch_compact(v, dy, dphi, want_iters=True)
    k1, k2, _ = s.last_kernel_times()
iters = int(it[0]); assert int(it.min()) == int(it.max())
wi, rb = s.last_lane_stats()
print(f"variant={os.environ.get('TPC_LANE_VARIANT','0')} H={H} {dtype} iters={iters} pg_ms={k2:.3f} "
      f"us_per_iter={k2*1e3/(iters-50):.4f} cycles@2.35GHz={k2*1e-3*2.35e9/(iters-50):.0f} wave_iters/wave={wi/1024:.0f} refills={rb}")
