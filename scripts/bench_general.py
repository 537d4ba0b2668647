#!/usr/bin/env python3
"""Diagnostic: general-form (per-instance A,B,C,Q,R,bounds,x0, per-step targets) throughput, fp64."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from trajectory_controller_amd import MpcSolver
from trajectory_controller_amd.synth import general_inputs

for I, H, n in ((2, 20, 262144), (2, 10, 262144), (1, 20, 262144), (2, 20, 4096)):
    g = general_inputs(H, n, I=I)
    names = ["A", "B", "C", "Q", "R", "lo", "hi", "x0", "targets"]
    dev = [torch.from_numpy(np.ascontiguousarray(g[k].reshape(n, -1).T)).cuda() for k in names]
    for algo in ("lane", "wave"):
        with MpcSolver(horizon=H, algo=algo) as s:
            s.set_profiling(True)
            for _ in range(2):
                u0, it = s.solve_batch_general(*dev, inputs=I, want_iters=True)
                k1, k2, _ = s.last_kernel_times()
        print(f"general I={I} H={H} n={n} {algo}: {k1 + k2:8.3f} ms  {n / (k1 + k2) / 1e3:8.3f} Msolve/s  mean iters {float(it.double().mean()):.0f}", flush=True)
