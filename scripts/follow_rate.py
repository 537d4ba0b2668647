#!/usr/bin/env python3
"""Diagnostic: tpc_mpc_follow_batch / _horizon on random polylines (24 points): scripts/follow_rate.py [n] [H]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from trajectory_controller_amd import MpcSolver
n = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
H = int(sys.argv[2]) if len(sys.argv) > 2 else 10
P = 24
rng = np.random.default_rng(11)
seg = rng.uniform(0.02, 0.25, size=(P, n)).astype(np.float32)
ang = np.cumsum(rng.uniform(-0.15, 0.15, size=(P, n)), axis=0).astype(np.float32)
px = np.cumsum(seg * np.cos(ang), axis=0, dtype=np.float32)
py = np.cumsum(seg * np.sin(ang), axis=0, dtype=np.float32)
g = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
args = [g(px), g(py), g(np.cos(ang).astype(np.float32)), g(np.sin(ang).astype(np.float32)),
        g(rng.uniform(0.6, 2.0, size=(P, n)).astype(np.float32)), g(np.full(n, P, dtype=np.int32)),
        g(rng.uniform(0.3, 4.0, size=n).astype(np.float32)), g(rng.uniform(0.2, 2.5, size=n).astype(np.float32))]
algo = sys.argv[3] if len(sys.argv) > 3 else "lane"
with MpcSolver(horizon=H, algo=algo) as s:
    for name, fn in (("follow_batch", s.follow_batch), ("follow_batch_horizon", s.follow_batch_horizon)):
        fn(*args)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            out = fn(*args)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5
        print(f"{name} n={n} H={H} P={P} {algo}: {dt * 1e3:.3f} ms per batch, {n / dt / 1e6:.2f} M cycles/s")
