#!/usr/bin/env python3
"""Diagnostic: throughput with 1, 2 or 3 batches in flight (one handle + one stream each)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from trajectory_controller_amd import MpcSolver
from trajectory_controller_amd.synth import compact_inputs
H = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = int(sys.argv[2]) if len(sys.argv) > 2 else 262144
steps = 12
v, dy, dphi = (torch.from_numpy(a).cuda() for a in compact_inputs(H, n))
for inflight in (1, 2, 3, 1, 2):
    solvers = [MpcSolver(horizon=H, algo="lane") for _ in range(inflight)]
    streams = [torch.cuda.Stream() for _ in range(inflight)]
    outs = [(torch.empty_like(v), torch.empty_like(v)) for _ in range(inflight)]
    for i in range(inflight):
        with torch.cuda.stream(streams[i]):
            solvers[i].solve_batch_compact(v, dy, dphi, out=outs[i], want_flags=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        i = k % inflight
        with torch.cuda.stream(streams[i]):
            solvers[i].solve_batch_compact(v, dy, dphi, out=outs[i], want_flags=False)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    same = all(torch.equal(outs[0][0], o[0]) and torch.equal(outs[0][1], o[1]) for o in outs)
    print(f"H={H} n={n} batches in flight {inflight}: {dt/steps*1e3:.3f} ms/step  {n*steps/dt/1e6:.2f} M solves/s  outputs identical {same}")
    for s in solvers: s.close()
