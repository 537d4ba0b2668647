#!/usr/bin/env python3
"""Diagnostic sweep: WAVE vs LANE kernel time over batch size and horizon (fp64, compact form)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from trajectory_controller_amd import MpcSolver
from trajectory_controller_amd.synth import compact_inputs

dtype = sys.argv[1] if len(sys.argv) > 1 else "f64"
tdt = torch.float64 if dtype == "f64" else torch.float32
print(f"{'H':>3} {'n':>8} | {'wave ms':>9} {'Msolve/s':>9} | {'lane ms':>9} {'Msolve/s':>9} | one-solve latency us")
for H in (4, 10, 20, 30):
    for n in (1, 64, 1024, 4096, 16384, 65536, 262144):
        v, dy, dphi = (torch.from_numpy(a).to("cuda", dtype=tdt) for a in compact_inputs(H, n))
        row = []
        for algo in ("wave", "lane"):
            with MpcSolver(horizon=H, dtype=dtype, algo=algo) as s:
                s.set_profiling(True)
                for _ in range(2):
                    s.solve_batch_compact(v, dy, dphi, want_flags=False)
                    k1, k2, _ = s.last_kernel_times()
                row.append(k1 + k2)
        lat = ""
        if n == 1:
            with MpcSolver(horizon=H, dtype="f64") as s:
                s.mpc_controller_tobi(1.0, 0.1, 0.05)
                t0 = time.perf_counter()
                for _ in range(50):
                    s.mpc_controller_tobi(1.0, 0.1, 0.05)
                lat = f"{(time.perf_counter() - t0) / 50 * 1e6:.1f}"
        print(f"{H:>3} {n:>8} | {row[0]:9.3f} {n / row[0] / 1e3:9.3f} | {row[1]:9.3f} {n / row[1] / 1e3:9.3f} | {lat}", flush=True)
