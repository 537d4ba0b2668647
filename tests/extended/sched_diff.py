#!/usr/bin/env python3
"""Differential check of the experimental LLVM scheduler used for the LANE units (csrc/Makefile):
dump the outputs of every LANE kernel (fp64 and fp32, compact and general, every horizon) to a file;
run once with the shipped library and once with TPC_MPC_LIB pointing at a build made with the default
scheduler, then compare the two dumps bit for bit.

    python tests/extended/sched_diff.py dump out.npz
    python tests/extended/sched_diff.py compare a.npz b.npz
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np


def dump(path):
    from trajectory_controller_amd import MpcSolver
    from trajectory_controller_amd.synth import compact_inputs, general_inputs
    out = {}
    soa = lambda a, n: np.ascontiguousarray(np.asarray(a).reshape(n, -1).T)
    for dtype in ("f64", "f32"):
        dt = np.float64 if dtype == "f64" else np.float32
        for H in (4, 5, 10, 20, 30, 40):
            n = 4096 if H <= 20 else 1024
            v, dy, dphi = compact_inputs(H, n, first=555)
            with MpcSolver(horizon=H, dtype=dtype, algo="lane") as s:
                f, r, it = s.solve_batch_compact(v.astype(dt), dy.astype(dt), dphi.astype(dt), want_iters=True)
            out[f"c_{dtype}_{H}_f"], out[f"c_{dtype}_{H}_r"], out[f"c_{dtype}_{H}_it"] = f, r, it
            for I in (1, 2):
                if H == 4:
                    continue
                m = 1024 if H <= 20 else 256
                g = general_inputs(H, m, I=I, first=99)
                with MpcSolver(horizon=H, dtype=dtype, algo="lane") as s:
                    u0, git = s.solve_batch_general(*[soa(g[k], m).astype(dt) for k in
                                                      ("A", "B", "C", "Q", "R", "lo", "hi", "x0", "targets")],
                                                    inputs=I, want_iters=True)
                out[f"g_{dtype}_{H}_{I}_u"], out[f"g_{dtype}_{H}_{I}_it"] = u0, git
            print("done", dtype, H, flush=True)
    np.savez(path, **out)


def compare(a, b):
    A, B = np.load(a), np.load(b)
    bad = 0
    for k in A.files:
        x, y = A[k], B[k]
        same = x.tobytes() == y.tobytes()
        bad += not same
        if not same:
            print("DIFFERENT:", k, float(np.nanmax(np.abs(x.astype(np.float64) - y.astype(np.float64)))))
    print(f"{len(A.files)} arrays compared, {bad} differ")
    return bad


if __name__ == "__main__":
    if sys.argv[1] == "dump":
        dump(sys.argv[2])
    else:
        sys.exit(1 if compare(sys.argv[2], sys.argv[3]) else 0)
