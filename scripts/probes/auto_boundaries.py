"""Diagnostic: AUTO around its crossovers (compact form, fp64 / fp32): which family ran, and the result against the
LANE family's on the same inputs (iteration counts, max |du|)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from trajectory_controller_amd import MpcSolver
from trajectory_controller_amd.synth import compact_inputs
NAMES = {1: "WAVE", 2: "LANE", 3: "LANE_FMA", 100: "generic"}
bad = 0
for dtype, tdt, tol in (("f64", torch.float64, 1e-9), ("f32", torch.float32, None)):
    for H in (4, 5, 10, 20, 30, 40):
        for n in (21503, 21504, 24575, 24576, 26623, 26624, 28671, 28672, 32768, 32769):
            v, dy, dphi = (torch.from_numpy(a).to("cuda", dtype=tdt) for a in compact_inputs(H, n, first=777))
            with MpcSolver(horizon=H, dtype=dtype, algo="lane") as s:
                lf, lr, lit = s.solve_batch_compact(v, dy, dphi, want_iters=True)
            with MpcSolver(horizon=H, dtype=dtype) as s:
                s.set_profiling(True)
                f, r, it = s.solve_batch_compact(v, dy, dphi, want_iters=True)
                fam = s.last_kernel_times()[2]
            same = float((it == lit).double().mean())
            err = float(torch.maximum((f - lf).abs().max(), (r - lr).abs().max()))
            ok = (tol is None) or (bool(torch.equal(it, lit)) and err <= tol)   # (a mean of ones need not be exactly 1.0)
            bad += not ok
            print(f"{dtype} N={H:2d} n={n:6d}: AUTO -> {NAMES.get(fam, fam):8s} iteration counts equal {same:.6f}  max|du| vs LANE {err:.2e}{'' if ok else '   <-- CHECK'}")
print("outside the fp64 tolerance:", bad)
