#!/usr/bin/env python3
"""Randomised differential run on the GPU box: LANE_FMA (fp64, or DTYPE=f32) against (a) the CPU MODEL of its own
arithmetic (tests/model/), BIT FOR BIT -- outputs, iteration counts, whichever stop-test build the kernels' screen
picks -- and (b) the CPU oracle (fp64 only): iteration counts and |du|.  Random weights, bounds (equal, unequal,
start point on a bound, start point outside the box, tiny), step size, wheelbase, eps, iteration caps, and a few
hostile speeds (0, 1e-12, 50, NaN, 1e70).  One line per parameter set; exits non-zero on any bit difference.
    [DTYPE=f32] python tests/extended/fuzz_lane_fma.py [sets] [n]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle.bindings import Oracle, build_oracle
from tests.model.bindings import UbModel
from trajectory_controller_amd import MpcSolver
from trajectory_controller_amd.synth import compact_inputs

build_oracle()
DT = os.environ.get("DTYPE", "f64")
orc, mdl = Oracle(dtype=DT), UbModel(DT)
NP, UI = (np.float64, np.uint64) if DT == "f64" else (np.float32, np.uint32)
rng = np.random.default_rng(int(os.environ.get("SEED", "20261005")))
sets = int(sys.argv[1]) if len(sys.argv) > 1 else 60
n = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
threads = int(os.environ.get("THREADS", "16"))
bad = flips = total = 0
worst = 0.0
for s_i in range(sets):
    # (fp64 at N = 30 / 40: GROUP takes LANE_FMA's requests since round 5 -- tests/extended/fuzz_group.py covers it;
    #  N = 20 with equal bounds is the hand-written kernel, csrc/mpc_ub_asm.h: weighted up)
    H = ((4, 5, 10, 20, 20, 20) if DT == "f64" else (4, 5, 10, 20, 30, 40))[s_i % 6]
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
    v, dy, dphi = v.astype(NP), (dy * scale).astype(NP), (dphi * scale).astype(NP)
    hostile = rng.random() < 0.3
    if hostile:
        v[rng.integers(0, m, 5)] = rng.choice([0.0, 1e-12, 50.0, np.nan, 1e70 if DT == "f64" else 1e30], 5)
    kw = dict(weights=w, T=T, l=l, lo=lo, hi=hi, eps=eps, max_iter=cap, smo_iters=smo, nthreads=threads)
    mf, mr, mit, _ = mdl.solve_compact(H, v, dy, dphi, fast_stop=None, **kw)
    of, orr, oit = orc.solve_compact(H, v, dy, dphi, **kw)
    with MpcSolver(horizon=H, algo="lane_fma", dtype=DT, weight_y=w[0], weight_phi=w[1], weight_steering_front=w[2],
                   weight_steering_rear=w[3], lower=lo, upper=hi, step_size=T, wheelbase=l, eps=eps,
                   max_iter=cap, smo_iters=smo) as s:
        f, r, it = s.solve_batch_compact(v, dy, dphi, want_iters=True)
    diff = lambda x, y: (x.view(UI) != y.view(UI)) & ~(np.isnan(x) & np.isnan(y))
    mism = int(np.sum(diff(f, mf) | diff(r, mr) | (it != mit)))
    bad += mism
    fin = np.isfinite(v) & (np.abs(T * v.astype(np.float64)) < (1e40 if DT == "f64" else 1e3))
    same = (it == oit) | ~fin
    err = np.where(fin & np.isfinite(of) & np.isfinite(orr), np.maximum(np.abs(f - of), np.abs(r - orr)), 0)
    done = same & (oit < cap)
    flips += int((~same).sum()); total += m
    if DT == "f64" and done.any():
        worst = max(worst, float(err[done].max()))
    print(f"set {s_i:3d} H={H:2d} n={m:5d} bounds kind {kind} eps {eps:.1e} cap {cap:5d} smo {smo:3d}{' hostile' if hostile else ''}: "
          f"vs model: {mism} bit differences; vs oracle: iteration counts differ on {int((~same).sum())}, "
          f"max|du| converged {err[done].max() if done.any() else 0:.2e}", flush=True)
print(f"{DT}: {sets} parameter sets, {total} instances: {bad} bit differences against the model; against the oracle "
      f"iteration counts differ on {flips} ({flips / total:.2e})" + (f", max |du| among the converged rest {worst:.2e}" if DT == "f64" else ""))
sys.exit(1 if bad else 0)
