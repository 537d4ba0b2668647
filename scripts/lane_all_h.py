#!/usr/bin/env python3
"""Diagnostic: LANE kernel times for every supported horizon at one batch size (fp64 and fp32)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from trajectory_controller_amd import MpcSolver
from trajectory_controller_amd.synth import compact_inputs
n = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
for dtype in ("f64", "f32"):
    tdt = torch.float64 if dtype == "f64" else torch.float32
    for H in (4, 5, 10, 20, 30, 40):
        m = n
        v, dy, dphi = (torch.from_numpy(a).to("cuda", dtype=tdt) for a in compact_inputs(H, m))
        with MpcSolver(horizon=H, dtype=dtype, algo="lane") as s:
            s.set_profiling(True)
            for _ in range(2):
                f, r, it = s.solve_batch_compact(v, dy, dphi, want_iters=True, want_flags=False)
                k1, k2, _ = s.last_kernel_times()
            wi, rb = s.last_lane_stats()
            pgit = float((it.double() - 50).clamp(min=0).sum())
            print(f"{dtype} H={H:2d} n={m:7d}: cd {k1:7.3f} ms  pg {k2:8.3f} ms  {m/(k1+k2)/1e3:8.2f} M solves/s  mean iters {float(it.double().mean()):7.1f}  "
                  f"us/wave-iter {k2*1e3/(wi/ (1024 if True else 1)) if wi else 0:6.3f} (x waves/SIMD)  util {pgit/(64.0*wi) if wi else 0:.3f}", flush=True)
