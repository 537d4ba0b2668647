#!/usr/bin/env python3
"""Large differential run on the GPU box: LANE (fp64) outputs against REAL dlib (oracle/_ref) on
fresh seeded instances, all horizons.  Prints one line per horizon; exits non-zero on any mismatch."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle.bindings import DlibRef, REF_SO
from trajectory_controller_amd import MpcSolver
from trajectory_controller_amd.synth import compact_inputs

ref = DlibRef(REF_SO)
threads = int(os.environ.get("THREADS", "16"))
bad = 0
for H, n in ((4, 1 << 20), (5, 1 << 20), (10, 1 << 20), (20, 1 << 20), (30, 1 << 18), (40, 1 << 17)):
    v, dy, dphi = compact_inputs(H, n, first=10_000_000)
    t0 = time.time()
    rf, rr = ref.solve_compact(H, v, dy, dphi, nthreads=threads)
    tc = time.time() - t0
    tv, ty, tp = (torch.from_numpy(a).cuda() for a in (v, dy, dphi))
    with MpcSolver(horizon=H, algo="lane") as s:
        f, r = s.solve_batch_compact(tv, ty, tp)
        torch.cuda.synchronize()
        t0 = time.time()
        f, r = s.solve_batch_compact(tv, ty, tp)
        torch.cuda.synchronize()
        tg = time.time() - t0
    f, r = f.cpu().numpy(), r.cpu().numpy()
    mism = int(np.sum((f != rf) | (r != rr)))
    bad += mism
    print(f"H={H:2d} n={n:8d}  mismatching instances: {mism}  max|du|={max(np.abs(f-rf).max(), np.abs(r-rr).max()):.3e}  "
          f"dlib {threads} threads {tc:6.1f} s ({n/tc/1e3:.1f} k/s)   GPU {tg*1e3:8.1f} ms ({n/tg/1e6:.1f} M/s)", flush=True)
sys.exit(1 if bad else 0)
