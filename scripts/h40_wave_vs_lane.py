#!/usr/bin/env python3
"""Diagnostic: the two-variables-per-lane WAVE kernel against LANE at N = 40 over batch size (the AUTO crossover)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from trajectory_controller_amd import MpcSolver
from trajectory_controller_amd.synth import compact_inputs
H = 40
for n in (2048, 4096, 8192, 12288, 16384, 24576, 32768):
    v, dy, dphi = (torch.from_numpy(a).cuda() for a in compact_inputs(H, n))
    row = []
    for algo in ("wave", "lane"):
        with MpcSolver(horizon=H, algo=algo) as s:
            s.set_profiling(True)
            best = 1e9
            for _ in range(2):
                s.solve_batch_compact(v, dy, dphi, want_flags=False)
                k1, k2, _ = s.last_kernel_times()
                best = min(best, k1 + k2)
        row.append(best)
    print(f"H={H} n={n:6d}: wave {row[0]:8.3f} ms  lane {row[1]:8.3f} ms", flush=True)
