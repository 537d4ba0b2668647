#!/usr/bin/env python3
"""Randomised differential run on the GPU box, general form: the GROUP kernels (csrc/mpc_groupg.h, every group size
built for the horizon, fp64) against the CPU oracle: iteration counts and |du|.  Per set: one or two inputs, N in
10, 20, 30, 40, random weights, per-instance bounds (symmetric, one-sided, pinned, huge -- the last three make the
screen refuse the batch: the one-lane family's exact kernels then solve it), random A off the compact structure,
disturbance C, target scale, eps, caps, warm starts with the controller state in and out every third set, and now and
then dead input columns, Q = 0 and non-finite entries.  Exits non-zero if an instance that did not end on the cap
differs from the oracle in its iteration count or by more than 1e-9.
    python tests/extended/fuzz_groupg.py [sets] [n]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle.bindings import Oracle, build_oracle
from trajectory_controller_amd import MpcSolver, capi
from trajectory_controller_amd.synth import general_inputs

build_oracle()
DT = "f64"
orc = Oracle(dtype=DT)
BUILT = {10: (2, 4), 20: (2, 4, 8), 30: (4, 8), 40: (4, 8)}
NP, UI = (np.float64, np.uint64) if DT == "f64" else (np.float32, np.uint32)
rng = np.random.default_rng(int(os.environ.get("SEED", "20261006")))
sets = int(sys.argv[1]) if len(sys.argv) > 1 else 48
n_all = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
threads = int(os.environ.get("THREADS", "16"))
G = ["A", "B", "C", "Q", "R", "lo", "hi", "x0", "targets"]
soa = lambda a: np.ascontiguousarray(np.asarray(a).reshape(a.shape[0], -1).T)
bad = flips = total = 0
worst = 0.0
for s_i in range(sets):
    H = (10, 20, 30, 40)[s_i % 4]
    I = 1 + (s_i // 4) % 2
    Gl = BUILT[H][(s_i // 8) % len(BUILT[H])]
    warm = s_i % 3 == 2
    n = n_all if H <= 20 else n_all // 4
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
    kw = {}
    cin = vin = None
    if warm:
        cin = rng.uniform(-0.3, 0.3, size=(n, H, I)); vin = rng.uniform(-0.3, 0.3, size=(n, H, I))
        ou0, cout, oit, vout = orc.solve_general(I, H, *args, eps=eps, max_iter=cap, smo_iters=smo, nthreads=threads, controls_in=cin, v_in=vin, want_v=True)
        kw = dict(controls=soa(cin), v_state=soa(vin))
    else:
        ou0, _, oit = orc.solve_general(I, H, *args, eps=eps, max_iter=cap, smo_iters=smo, nthreads=threads)
    with MpcSolver(horizon=H, algo="group", dtype=DT, eps=eps, max_iter=cap, smo_iters=smo) as s:
        s.set_option(capi.OPT_GROUP_LANES, Gl)
        s.set_profiling(True)
        u0, it = s.solve_batch_general(*[soa(a) for a in args], inputs=I, want_iters=True, **kw)
        assert s.last_kernel_times()[2] == 4, "GROUP did not run"
    u0 = u0.T
    mism = 0
    fin = np.isfinite(g["x0"]).all(axis=1) & (np.abs(g["x0"]) < 1e20).all(axis=1)
    same = (it == oit) | ~fin
    done = same & (oit < cap) & fin
    flips += int((~same).sum()); total += n
    err = float(np.nanmax(np.abs(u0[done] - ou0[done]))) if done.any() else 0.0
    if warm and done.any():
        err = max(err, float(np.nanmax(np.abs(kw["controls"].T.reshape(n, H, I)[done] - cout[done]))), float(np.nanmax(np.abs(kw["v_state"].T.reshape(n, H, I)[done] - vout[done]))))
    worst = max(worst, err)
    bad += int((~same & fin & (oit < cap)).sum()) + (1 if err > 1e-9 else 0)
    print(f"set {s_i:3d} I={I} H={H:2d} G={Gl} n={n} bounds kind {kind} eps {eps:.1e} cap {cap:5d} smo {smo:3d}{' warm' if warm else ''}{' hostile' if hostile else ''}: "
          f"iteration counts differ from the oracle's on {int((~same).sum())}, max|du| converged {err:.2e}", flush=True)
print(f"general form, GROUP: {sets} parameter sets, {total} instances: iteration counts differ from the oracle's on {flips} "
      f"({flips / max(total, 1):.2e}), max |du| (controls and v of warm-started sets included) among the converged rest {worst:.2e}")
sys.exit(1 if bad else 0)
