#!/usr/bin/env python3
"""The PCIe-inclusive rate of the headline workload: inputs and outputs in HOST memory (numpy arrays through
TPC_MPC_HOST: the library stages them), against the same batch with arrays resident in HBM.
    python scripts/host_staging_rate.py [n] [H]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from trajectory_controller_amd import MpcSolver
from trajectory_controller_amd.synth import compact_inputs
n = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
H = int(sys.argv[2]) if len(sys.argv) > 2 else 20
v, dy, dphi = compact_inputs(H, n)
tv, ty, tp = (torch.from_numpy(a).cuda() for a in (v, dy, dphi))
with MpcSolver(horizon=H) as s:
    s.reserve(n)
    for name, args in (("device arrays", (tv, ty, tp)), ("host arrays (numpy, pageable)", (v, dy, dphi))):
        for _ in range(3):
            s.solve_batch_compact(*args, want_flags=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        K = 20
        for _ in range(K):
            f, r = s.solve_batch_compact(*args, want_flags=False)[:2]
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / K
        print(f"H={H} n={n} {name}: {dt * 1e3:.3f} ms per batch, {n / dt / 1e6:.2f} M solves/s", flush=True)
