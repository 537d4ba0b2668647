#!/usr/bin/env python3
"""Diagnostic: WAVE vs LANE vs LANE_FMA kernel time around the AUTO crossover (compact form).
    python scripts/crossover.py [f64|f32] [H ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from trajectory_controller_amd import MpcSolver
from trajectory_controller_amd.synth import compact_inputs
args = sys.argv[1:]
dtype = args.pop(0) if args and args[0] in ("f64", "f32") else "f64"
tdt = torch.float64 if dtype == "f64" else torch.float32
for H in [int(a) for a in args] or [4, 5, 10, 20, 30, 40]:
    for n in (2048, 4096, 8192, 12288, 16384, 24576, 32768, 49152, 65536):
        v, dy, dphi = (torch.from_numpy(a).to('cuda', dtype=tdt) for a in compact_inputs(H, n))
        row = []
        for algo in ("wave", "lane", "lane_fma"):
            with MpcSolver(horizon=H, algo=algo, dtype=dtype) as s:
                s.set_profiling(True)
                best = 1e9
                for _ in range(3):
                    s.solve_batch_compact(v, dy, dphi, want_flags=False)
                    k1, k2, _ = s.last_kernel_times()
                    best = min(best, k1 + k2)
                row.append(best)
        print(f"{dtype} H={H:2d} n={n:6d}: wave {row[0]:7.3f} ms  lane {row[1]:7.3f} ms  lane_fma {row[2]:7.3f} ms  -> {('WAVE', 'LANE', 'LANE_FMA')[int(np.argmin(row))]}", flush=True)
