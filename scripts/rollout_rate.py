#!/usr/bin/env python3
"""Diagnostic: closed-loop rollout rate (warm-started controllers, general form built from the compact model)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from trajectory_controller_amd import MpcSolver
from trajectory_controller_amd.synth import general_inputs
H = int(sys.argv[1]) if len(sys.argv) > 1 else 10
n = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
g = general_inputs(H, n, I=2)
soa = lambda a: torch.from_numpy(np.ascontiguousarray(np.asarray(a).reshape(n, -1).T)).cuda()
dev = [soa(g[k]) for k in ("A", "B", "C", "Q", "R", "lo", "hi", "x0", "targets")]
for algo in (sys.argv[4].split(",") if len(sys.argv) > 4 else ("lane", "wave", "group", "auto")):
    with MpcSolver(horizon=H, algo=algo) as s:
        s.rollout(2, *dev, inputs=2, want_iters=True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        c, st, it = s.rollout(steps, *dev, inputs=2, want_iters=True)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        it = it.double()
        print(f"H={H} n={n} steps={steps} {algo}: {dt*1e3:.1f} ms  {n*steps/dt/1e6:.2f} M controller-steps/s  "
              f"mean iterations: first step {float(it[0].mean()):.0f}, later steps {float(it[1:].mean()):.0f}")
