#!/usr/bin/env python3
"""Diagnostic: end-to-end latency of one mpcControllerTobi call (tpc_mpc_solve_one) per horizon."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from trajectory_controller_amd import MpcSolver
for H in (4, 10, 20, 40):
    with MpcSolver(horizon=H) as s:
        for _ in range(20):
            s.mpc_controller_tobi(1.0, 0.1, 0.05)
        ts = []
        for i in range(300):
            t0 = time.perf_counter()
            f, r = s.mpc_controller_tobi(1.0 + 1e-3 * i, 0.1, 0.05)
            ts.append(time.perf_counter() - t0)
        ts = np.array(ts) * 1e6
        print(f"H={H:2d}: median {np.median(ts):6.1f} us  p10 {np.percentile(ts,10):6.1f}  p90 {np.percentile(ts,90):6.1f}  (last result {f:+.6f} {r:+.6f})")
