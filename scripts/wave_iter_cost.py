#!/usr/bin/env python3
"""Diagnostic: cost of one WAVE iteration (coordinate-descent and projected-gradient phase), per wavefront
and per SIMD, by batch size.  Every instance is forced through exactly max_iter iterations (a small eps, caps
of 50 and 150); the difference between two caps isolates a loop.  usage: wave_iter_cost.py H [n ...]   (TPC_MPC_LIB selects the build)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from trajectory_controller_amd import MpcSolver
from trajectory_controller_amd.synth import compact_inputs
H = int(sys.argv[1])
sizes = [int(a) for a in sys.argv[2:]] or [256, 1024, 2048, 4096]
tag = os.path.basename(os.path.dirname(os.environ.get("TPC_MPC_LIB", "/shipped/x")))
for n in sizes:
    v, dy, dphi = (torch.from_numpy(a).cuda() for a in compact_inputs(H, n))
    # eps small enough that no instance converges inside the caps, large enough for the arithmetic mask
    # (the screen wants lambda * max|bound| * 2^-50 < eps): the loops the reference workload runs
    def run(**kw):
        with MpcSolver(horizon=H, algo="wave", eps=1e-10, **kw) as s:
            s.set_profiling(True)
            ts = []
            for _ in range(4):
                _, _, it = s.solve_batch_compact(v, dy, dphi, want_iters=True, want_flags=False)
                ts.append(s.last_kernel_times()[0])
        return min(ts), float((it < kw["max_iter"]).double().mean())
    t0, _ = run(max_iter=0)
    t50, e0 = run(max_iter=50)
    t_pg, e1 = run(max_iter=150)
    t_cd, e2 = run(max_iter=150, smo_iters=150)
    pg = 1e6 * (t_pg - t50) / 100
    cd = 1e6 * (t_cd - t50) / 100
    w = max(1, n / 1024)
    print(f"[{tag}] H={H} n={n}: launch + prologue {t0*1e3:.1f} us; per iteration and wave: CD {cd:.1f} ns, PG {pg:.1f} ns; "
          f"per SIMD: CD {cd / w:.1f} ns, PG {pg / w:.1f} ns   (stopped before the cap: {max(e0, e1, e2):.3f})", flush=True)
