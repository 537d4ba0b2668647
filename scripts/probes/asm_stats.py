#!/usr/bin/env python3
"""Headline batch through LANE_FMA: PG kernel time, wave iterations and refill passes per wavefront (stats words)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from trajectory_controller_amd import MpcSolver
from trajectory_controller_amd.synth import compact_inputs
H, n = 20, int(sys.argv[1]) if len(sys.argv) > 1 else 262144
v, dy, dphi = (torch.from_numpy(a).cuda() for a in compact_inputs(H, n))
s = MpcSolver(horizon=H, dtype="f64", algo="lane_fma")
s.set_profiling(True)
for rep in range(4):
    f, r, it = s.solve_batch_compact(v, dy, dphi, want_iters=True)
    k1, k2, _ = s.last_kernel_times()
    wi, rb = s.last_lane_stats()
    print(f"n={n} cd_ms={k1:.3f} pg_ms={k2:.3f} wave_iters/wave={wi/1024:.1f} refills/wave={rb/1024:.1f} "
          f"us_per_wave_iter={k2*1e3/(wi/1024):.4f} mean_iters={float(it.double().mean()):.1f}")
