#!/usr/bin/env python3
"""Randomised differential run on the GPU box: the GROUP family (G lanes per instance, every size built for the
horizon, fp64) against the CPU oracle -- iteration counts and |du| -- and against the LANE_FMA kernels (the same
arithmetic in another association: iteration counts).  Random weights, bounds (equal, unequal, start point on a bound,
start point outside the box -> the exact build of LANE_FMA takes the batch, tiny), step size, wheelbase, eps,
iteration caps, and a few hostile speeds (0, 1e-12, 50, NaN, 1e70).  One line per parameter set; exits non-zero if an
iteration count differs from the oracle's on an instance that did not end on the cap, or |du| exceeds 1e-9 there.
    python tests/extended/fuzz_group.py [sets] [n]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle.bindings import Oracle, build_oracle
from trajectory_controller_amd import MpcSolver, capi
from trajectory_controller_amd.synth import compact_inputs

build_oracle()
orc = Oracle()
rng = np.random.default_rng(int(os.environ.get("SEED", "20261006")))
sets = int(sys.argv[1]) if len(sys.argv) > 1 else 60
n = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
threads = int(os.environ.get("THREADS", "16"))
BUILT = {10: (2, 4), 20: (2, 4, 8), 30: (2, 4, 8), 40: (2, 4, 8)}
flips = total = bad = 0
worst = worst_cap = 0.0
for s_i in range(sets):
    H = (10, 20, 30, 40)[s_i % 4]
    G = BUILT[H][(s_i // 4) % len(BUILT[H])]
    m = n if H <= 20 else n // 4
    w = (10 ** rng.uniform(-1, 2), 10 ** rng.uniform(-1, 2), 10 ** rng.uniform(-4, 1), 10 ** rng.uniform(-2, 1.5))
    kind = rng.integers(0, 5)
    if kind == 0:   lo, hi = (-0.384, -0.384), (0.384, 0.384)
    elif kind == 1: a, b = rng.uniform(0.02, 0.6, 2); lo, hi = (-a, -b), (b, a)
    elif kind == 2: a = rng.uniform(0.05, 0.5); lo, hi = (0.0, -a), (a, 0.0)
    elif kind == 3: a = rng.uniform(0.05, 0.3); lo, hi = (a / 2, -a), (a, -a / 2)
    else:           a = rng.uniform(1e-3, 2e-2); lo, hi = (-a, -a), (a, a)
    T = float(rng.uniform(0.02, 0.3)); l = float(rng.uniform(0.1, 0.5))
    eps = float(10 ** rng.uniform(-4, -1)); cap = int(rng.choice([10000, 10000, 10000, 300, 77, 51, 50, 20]))
    smo = int(rng.choice([50, 50, 50, 0, 7, 120]))
    v, dy, dphi = compact_inputs(H, m, first=int(rng.integers(0, 1 << 30)))
    scale = float(rng.choice([1.0, 1.0, 0.2, 3.0]))
    dy, dphi = dy * scale, dphi * scale
    hostile = rng.random() < 0.3
    if hostile:
        v[rng.integers(0, m, 5)] = rng.choice([0.0, 1e-12, 50.0, np.nan, 1e70], 5)
    kw = dict(weights=w, T=T, l=l, lo=lo, hi=hi, eps=eps, max_iter=cap, smo_iters=smo, nthreads=threads)
    of, orr, oit = orc.solve_compact(H, v, dy, dphi, **kw)
    skw = dict(weight_y=w[0], weight_phi=w[1], weight_steering_front=w[2], weight_steering_rear=w[3], lower=lo, upper=hi,
               step_size=T, wheelbase=l, eps=eps, max_iter=cap, smo_iters=smo)
    with MpcSolver(horizon=H, algo="group", **skw) as s:
        s.set_option(capi.OPT_GROUP_LANES, G)
        f, r, it = s.solve_batch_compact(v, dy, dphi, want_iters=True)
    with MpcSolver(horizon=H, algo="lane_fma", **skw) as s:
        lf, lr, lit = s.solve_batch_compact(v, dy, dphi, want_iters=True)
    fin = np.isfinite(v) & (np.abs(T * v) < 1e40)
    same = (it == oit) | ~fin
    err = np.where(fin & np.isfinite(of) & np.isfinite(orr), np.maximum(np.abs(f - of), np.abs(r - orr)), 0)
    done = fin & (oit < cap)
    capped = fin & (oit == cap)
    flips += int((~same).sum()); total += m
    if done.any(): worst = max(worst, float(err[done].max()))
    if capped.any(): worst_cap = max(worst_cap, float(err[capped].max()))
    bad += int((~same & done).sum()) + int((err[done] > 1e-9).sum() if done.any() else 0)
    print(f"set {s_i:3d} H={H:2d} G={G} n={m:5d} bounds kind {kind} eps {eps:.1e} cap {cap:5d} smo {smo:3d}{' hostile' if hostile else ''}: "
          f"iteration counts differ from the oracle's on {int((~same).sum())}, from LANE_FMA's on {int(((it != lit) & fin).sum())}; "
          f"max|du| converged {err[done].max() if done.any() else 0:.2e}, cut off by the cap ({int(capped.sum())}) {err[capped].max() if capped.any() else 0:.2e}", flush=True)
print(f"{sets} parameter sets, {total} instances: iteration counts differ from the oracle's on {flips} ({flips / total:.2e}); "
      f"max |du| among converged instances {worst:.2e}, among instances cut off by the cap {worst_cap:.2e}")
sys.exit(1 if bad else 0)
