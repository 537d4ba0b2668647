#!/usr/bin/env python3
"""Diagnostic: WAVE kernel time over batch size and horizon (fp64, compact form)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from trajectory_controller_amd import MpcSolver
from trajectory_controller_amd.synth import compact_inputs
for H in (4, 10, 20):
    for n in (1, 4096, 16384, 262144):
        v, dy, dphi = (torch.from_numpy(a).cuda() for a in compact_inputs(H, n))
        with MpcSolver(horizon=H, algo="wave") as s:
            s.set_profiling(True)
            for _ in range(2):
                s.solve_batch_compact(v, dy, dphi, want_flags=False)
                k1, k2, _ = s.last_kernel_times()
        print(f"wave H={H:2d} n={n:7d}  {k1:9.3f} ms  {n / k1 / 1e3:8.3f} Msolve/s", flush=True)
