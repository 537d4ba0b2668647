#!/usr/bin/env python3
"""Diagnostic (needs a library built with -DTPC_ONE_TIMING: scripts/build_one_variant.sh timing "-DTPC_ONE_TIMING",
then TPC_MPC_LIB=ab/timing/libtpc_mpc.so): where a resident single solve spends its time, per horizon and kind of request.
The library prints the device-side split (shader clocks) when the handle is destroyed; this prints the host-side mean."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trajectory_controller_amd import MpcSolver
for H in [int(a) for a in sys.argv[1:]] or [4, 10]:
    for kind, vary_v, cap in (("new v", True, 10000), ("same v", False, 10000), ("new v, max_iter=0", True, 0), ("same v, max_iter=0", False, 0)):
        with MpcSolver(horizon=H, max_iter=cap) as s:
            s.solve_one(1.0, 0.1, 0.05)
            n = 3000
            t0 = time.perf_counter()
            for i in range(n):
                v = 0.5 + 3.0 * ((i * 2654435761) % 1000) / 1000.0 if vary_v else 1.7
                s.solve_one(v, 0.1 + 1e-4 * (i % 97), -0.2 + 4e-3 * (i % 89))
            dt = (time.perf_counter() - t0) / n
            print(f"N={H} {kind}: host mean {dt * 1e6:.2f} us (python call overhead included)", file=sys.stderr, flush=True)
