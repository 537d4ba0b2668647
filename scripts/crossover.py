#!/usr/bin/env python3
"""Diagnostic: WAVE vs LANE kernel time around the AUTO crossover (fp64, compact form)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from trajectory_controller_amd import MpcSolver
from trajectory_controller_amd.synth import compact_inputs
for H in [int(a) for a in sys.argv[1:]] or [4, 10, 20]:
    for n in (2048, 4096, 8192, 12288, 16384, 24576, 32768, 49152, 65536):
        v, dy, dphi = (torch.from_numpy(a).cuda() for a in compact_inputs(H, n))
        row = []
        for algo in ("wave", "lane"):
            with MpcSolver(horizon=H, algo=algo) as s:
                s.set_profiling(True)
                best = 1e9
                for _ in range(3):
                    s.solve_batch_compact(v, dy, dphi, want_flags=False)
                    k1, k2, _ = s.last_kernel_times()
                    best = min(best, k1 + k2)
                row.append(best)
        print(f"H={H:2d} n={n:6d}: wave {row[0]:7.3f} ms  lane {row[1]:7.3f} ms  -> {'WAVE' if row[0] < row[1] else 'LANE'}", flush=True)
