#!/usr/bin/env python3
"""Diagnostic: general-form throughput of one shape: scripts/general_rate.py I H [algo] [n] [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from trajectory_controller_amd import MpcSolver
from trajectory_controller_amd.synth import general_inputs
I, H = int(sys.argv[1]), int(sys.argv[2])
algo = sys.argv[3] if len(sys.argv) > 3 else "lane"
n = int(sys.argv[4]) if len(sys.argv) > 4 else 262144
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 4
g = general_inputs(H, n, I=I)
names = ["A", "B", "C", "Q", "R", "lo", "hi", "x0", "targets"]
dev = [torch.from_numpy(np.ascontiguousarray(g[k].reshape(n, -1).T)).cuda() for k in names]
with MpcSolver(horizon=H, algo=algo) as s:
    s.set_profiling(True)
    for _ in range(reps):
        u0, it = s.solve_batch_general(*dev, inputs=I, want_iters=True)
        k1, k2, _ = s.last_kernel_times()
print(f"general I={I} H={H} n={n} {algo}: {k1 + k2:8.3f} ms (cd {k1:.3f} pg {k2:.3f})  {n / (k1 + k2) / 1e3:8.3f} M solves/s  mean iters {float(it.double().mean()):.0f}")
