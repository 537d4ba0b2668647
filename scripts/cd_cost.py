#!/usr/bin/env python3
"""Diagnostic: what the coordinate-descent kernel costs per iteration and outside its loop."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from trajectory_controller_amd import MpcSolver
from trajectory_controller_amd.synth import compact_inputs
H = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 262144
v, dy, dphi = (torch.from_numpy(a).cuda() for a in compact_inputs(H, n))
for smo in (0, 10, 25, 50):
    with MpcSolver(horizon=H, algo="lane", smo_iters=smo, max_iter=smo) as s:   # cap = smo: the PG kernel only publishes
        s.set_profiling(True)
        best = 1e9
        for _ in range(3):
            s.solve_batch_compact(v, dy, dphi, want_flags=False)
            k1, k2, _ = s.last_kernel_times()
            best = min(best, k1)
        print(f"H={H} smo_iters={smo:2d}: CD + sort {best*1e3:7.1f} us   (PG publish-only pass {k2*1e3:7.1f} us)")
