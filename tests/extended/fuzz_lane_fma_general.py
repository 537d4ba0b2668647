#!/usr/bin/env python3
"""Randomised differential run on the GPU box, general form: LANE_FMA (csrc/mpc_ubg.h; fp64, or DTYPE=f32) against
(a) the CPU MODEL of its arithmetic (tests/model/), BIT FOR BIT -- outputs, iteration counts, whichever stop-test
build the screen picks -- and (b) the CPU oracle (fp64): iteration counts and |du|.  Per set: one or two inputs, N in
4, 5, 10, 20, random weights, per-instance bounds (symmetric, one-sided, pinned, huge), random A off the compact
structure, disturbance C, target scale, eps, caps, and now and then dead input columns, Q = 0 and non-finite entries.
    [DTYPE=f32] python tests/extended/fuzz_lane_fma_general.py [sets] [n]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle.bindings import Oracle, build_oracle
from tests.model.bindings import UbModel
from trajectory_controller_amd import MpcSolver
from trajectory_controller_amd.synth import general_inputs

build_oracle()
DT = os.environ.get("DTYPE", "f64")
orc, mdl = Oracle(dtype=DT), UbModel(DT)
NP, UI = (np.float64, np.uint64) if DT == "f64" else (np.float32, np.uint32)
rng = np.random.default_rng(int(os.environ.get("SEED", "20261006")))
sets = int(sys.argv[1]) if len(sys.argv) > 1 else 48
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
threads = int(os.environ.get("THREADS", "16"))
G = ["A", "B", "C", "Q", "R", "lo", "hi", "x0", "targets"]
soa = lambda a: np.ascontiguousarray(np.asarray(a).reshape(a.shape[0], -1).T)
bad = flips = total = 0
worst = 0.0
for s_i in range(sets):
    H = (4, 5, 10, 20)[s_i % 4]
    I = 1 + (s_i // 4) % 2
    g = general_inputs(H, n, I=I, first=int(rng.integers(0, 1 << 30)))
    g["Q"] = g["Q"] * 10 ** rng.uniform(-1.5, 1.0, size=(1, 2))
    g["R"] = g["R"] * 10 ** rng.uniform(-1.0, 1.5, size=(1, I))
    g["A"] = g["A"] + rng.normal(0, 0.02, size=g["A"].shape) * rng.choice([0.0, 1.0])
    g["targets"] = g["targets"] * float(rng.choice([1.0, 0.2, 5.0]))
    kind = int(rng.integers(0, 5))
    if kind == 1:   g["lo"] = -rng.uniform(0.02, 0.6, size=g["lo"].shape); g["hi"] = rng.uniform(0.02, 0.6, size=g["hi"].shape)
    elif kind == 2: g["lo"][::3] = 0.0                                   # the start point on a bound
    elif kind == 3: g["hi"][::5] = g["lo"][::5]; g["lo"][1::5] = 0.05    # pinned; 0 outside the box
    elif kind == 4: g["lo"][:] = -1e6; g["hi"][::2] = np.inf             # wide and one-sided-infinite boxes
    hostile = rng.random() < 0.35
    if hostile:
        g["B"][rng.integers(0, n, 4), 0] = 0.0
        g["Q"][rng.integers(0, n, 4)] = 0.0
        g["x0"][rng.integers(0, n, 2), 0] = rng.choice([np.nan, 1e30 if DT == "f32" else 1e200])
    g = {k: np.ascontiguousarray(a, dtype=NP) for k, a in g.items()}
    eps = float(10 ** rng.uniform(-4, -1)); cap = int(rng.choice([10000, 10000, 10000, 300, 77, 51, 50, 20]))
    smo = int(rng.choice([50, 50, 50, 0, 7, 120]))
    args = [g[k] for k in G]
    mu0, mit, _ = mdl.solve_general(I, H, *args, eps=eps, max_iter=cap, smo_iters=smo, nthreads=threads)
    ou0, _, oit = orc.solve_general(I, H, *args, eps=eps, max_iter=cap, smo_iters=smo, nthreads=threads)
    with MpcSolver(horizon=H, algo="lane_fma", dtype=DT, eps=eps, max_iter=cap, smo_iters=smo) as s:
        s.set_profiling(True)
        u0, it = s.solve_batch_general(*[soa(a) for a in args], inputs=I, want_iters=True)
        assert s.last_kernel_times()[2] == 3, "LANE_FMA did not run"
    u0 = u0.T
    diff = lambda x, y: (x.view(UI) != y.view(UI)) & ~(np.isnan(x) & np.isnan(y))
    mism = int(np.sum(diff(np.ascontiguousarray(u0), np.ascontiguousarray(mu0)).any(axis=1) | (it != mit)))
    bad += mism
    fin = np.isfinite(g["x0"]).all(axis=1) & (np.abs(g["x0"]) < 1e20).all(axis=1)
    same = (it == oit) | ~fin
    done = same & (oit < cap) & fin
    flips += int((~same).sum()); total += n
    err = float(np.nanmax(np.abs(u0[done] - ou0[done]))) if DT == "f64" and done.any() else 0.0
    worst = max(worst, err)
    print(f"set {s_i:3d} I={I} H={H:2d} n={n} bounds kind {kind} eps {eps:.1e} cap {cap:5d} smo {smo:3d}{' hostile' if hostile else ''}: "
          f"vs model: {mism} bit differences; vs oracle: iteration counts differ on {int((~same).sum())}, max|du| converged {err:.2e}", flush=True)
print(f"{DT} general form: {sets} parameter sets, {total} instances: {bad} bit differences against the model; against the oracle "
      f"iteration counts differ on {flips} ({flips / max(total, 1):.2e}), max |du| among the converged rest {worst:.2e}")
sys.exit(1 if bad else 0)
