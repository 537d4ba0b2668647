#!/usr/bin/env python3
"""Diagnostic: the general-form GROUP kernels -- parity against the oracle and kernel times beside WAVE / LANE_FMA / LANE.
    python scripts/groupg_probe.py I H G[,G...] n[,n...] [check]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from trajectory_controller_amd import MpcSolver, capi
from trajectory_controller_amd.synth import general_inputs
I, H = int(sys.argv[1]), int(sys.argv[2])
Gs = [int(x) for x in sys.argv[3].split(",")]; ns = [int(x) for x in sys.argv[4].split(",")]
check = len(sys.argv) > 5 and sys.argv[5] == "check"
names = ["A", "B", "C", "Q", "R", "lo", "hi", "x0", "targets"]
for n in ns:
    g = general_inputs(H, n, I=I, first=77000)
    dev = [torch.from_numpy(np.ascontiguousarray(g[k].reshape(n, -1).T)).cuda() for k in names]
    ref = None
    if check:
        from oracle.bindings import Oracle
        ref = Oracle().solve_general(I, H, *[g[k] for k in names], nthreads=16)
    for algo, G in [("wave", 0), ("lane_fma", 0), ("lane", 0)] + [("group", G) for G in Gs]:
        if algo == "wave" and (n > 32768 or I * H > 64):
            continue
        with MpcSolver(horizon=H, algo=algo) as s:
            if G:
                s.set_option(capi.OPT_GROUP_LANES, G)
            s.set_profiling(True)
            best = 1e9
            for _ in range(3):
                u0, it = s.solve_batch_general(*dev, inputs=I, want_iters=True)
                k1, k2, ran = s.last_kernel_times()
                best = min(best, k1 + k2)
        msg = f"general I={I} H={H} n={n:6d} {algo:8s} G={G} ran={ran}: {best:8.3f} ms"
        if ref is not None:
            u = u0.cpu().numpy().T
            msg += f"  max|du| {np.abs(u - ref[0]).max():.2e} iters equal {np.mean(it.cpu().numpy() == ref[2]):.6f}"
        print(msg, flush=True)
