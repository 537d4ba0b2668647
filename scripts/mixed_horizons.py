#!/usr/bin/env python3
"""Diagnostic: BASELINE config 5 -- 65 536 instances split evenly over N in {5, 10, 20, 40}, one call."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from trajectory_controller_amd import MpcSolver
from trajectory_controller_amd.synth import compact_inputs
Hs, per = (5, 10, 20, 40), 16384
parts = [compact_inputs(H, per) for H in Hs]
v, dy, dphi = (np.concatenate([p[c] for p in parts]) for c in range(3))
hz = np.repeat(np.array(Hs), per)
perm = np.random.default_rng(3).permutation(len(hz))
from trajectory_controller_amd import capi
for dtype in (sys.argv[1:] or ["f64", "f64fast", "f32"]):
    fast = dtype.endswith("fast")   # TPC_MPC_PARAM_FAST_CAPPED: capped instances keep the tolerance family's answer
    label, dtype = dtype, dtype[:3]
    tdt = torch.float64 if dtype == "f64" else torch.float32
    tv, ty, tp = (torch.from_numpy(a[perm]).to("cuda", dtype=tdt) for a in (v, dy, dphi))
    with MpcSolver(horizon=20, dtype=dtype, options=capi.PARAM_FAST_CAPPED if fast else 0) as s:
        s.solve_batch_compact_mixed(hz[perm], tv, ty, tp)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3):
            s.solve_batch_compact_mixed(hz[perm], tv, ty, tp)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    print(f"{label}: {dt*1e3:.1f} ms per mixed batch of {len(hz)}  -> {len(hz)/dt/1e6:.2f} M solves/s")
