#!/usr/bin/env python3
"""Diagnostic: what a perfect work hint (previous cycle's iteration counts) buys on the bench workload."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from trajectory_controller_amd import MpcSolver
from trajectory_controller_amd.synth import compact_inputs
H = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = int(sys.argv[2]) if len(sys.argv) > 2 else 262144
drift = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
v, dy, dphi = (torch.from_numpy(a).cuda() for a in compact_inputs(H, n))
s = MpcSolver(horizon=H, algo="lane"); s.set_profiling(True)
f, r, it = s.solve_batch_compact(v, dy, dphi, want_iters=True)
for rep in range(3):
    f, r, it = s.solve_batch_compact(v, dy, dphi, want_iters=True)
    k1, k2, _ = s.last_kernel_times(); wi, rb = s.last_lane_stats()
    print(f"H={H} n={n} lambda order : cd {k1:.3f} ms pg {k2:.3f} ms  wave_iters {wi} refills {rb}  -> {n/(k1+k2)/1e3:.2f} M/s")
g = torch.Generator(device="cuda").manual_seed(1)
for rep in range(4):
    if drift > 0:   # next control cycle: every input moves a little
        v = (v + drift * 3.9 * (torch.rand(n, generator=g, device="cuda", dtype=torch.float64) - 0.5)).clamp(0.1, 4.0)
        dy = dy + drift * (torch.rand(n, generator=g, device="cuda", dtype=torch.float64) - 0.5)
        dphi = dphi + drift * 1.2 * (torch.rand(n, generator=g, device="cuda", dtype=torch.float64) - 0.5)
    s.set_work_hint(it)
    f, r, it = s.solve_batch_compact(v, dy, dphi, want_iters=True)
    k1, k2, _ = s.last_kernel_times(); wi, rb = s.last_lane_stats()
    print(f"H={H} n={n} hinted (drift {drift}): cd {k1:.3f} ms pg {k2:.3f} ms  wave_iters {wi} refills {rb}  -> {n/(k1+k2)/1e3:.2f} M/s")
