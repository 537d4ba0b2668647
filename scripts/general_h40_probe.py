#!/usr/bin/env python3
"""Diagnostic: one small general-form batch at H = 40 (I given on the command line), LANE, vs nothing (just runs)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from trajectory_controller_amd import MpcSolver
from trajectory_controller_amd.synth import general_inputs
I = int(sys.argv[1]); H = int(sys.argv[2]) if len(sys.argv) > 2 else 40; n = 128
g = general_inputs(H, n, I=I)
soa = lambda a: np.ascontiguousarray(np.asarray(a).reshape(n, -1).T)
print("calling", I, H, flush=True)
with MpcSolver(horizon=H, algo="lane") as s:
    u0, it = s.solve_batch_general(*[soa(g[k]) for k in ("A", "B", "C", "Q", "R", "lo", "hi", "x0", "targets")], inputs=I, want_iters=True)
print("ok", I, H, float(np.abs(u0).max()), int(it.max()), flush=True)
