#!/usr/bin/env python3
"""Diagnostic: WAVE work-queue kernel time by horizon and batch size (fp64, compact form), best of 4.
usage: wave_queue_times.py [H ...]   (TPC_MPC_LIB selects the build, e.g. an A/B variant from build_wave_variant.sh)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from trajectory_controller_amd import MpcSolver
from trajectory_controller_amd.synth import compact_inputs
tag = os.path.basename(os.path.dirname(os.environ.get("TPC_MPC_LIB", "/shipped/x")))
for H in [int(a) for a in sys.argv[1:]] or [10, 20, 30]:
    for n in (4096, 8192, 16384, 32768) if H >= 10 else (4096, 16384, 32768):
        v, dy, dphi = (torch.from_numpy(a).cuda() for a in compact_inputs(H, n))
        with MpcSolver(horizon=H, algo="wave") as s:
            s.set_profiling(True)
            ts = []
            for _ in range(4):
                s.solve_batch_compact(v, dy, dphi, want_flags=False)
                ts.append(s.last_kernel_times()[0])
        print(f"[{tag}] wave H={H:2d} n={n:6d}  {min(ts) * 1e3:9.1f} us  {n / min(ts) / 1e3:8.3f} M solves/s", flush=True)
