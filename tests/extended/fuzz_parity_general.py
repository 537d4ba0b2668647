#!/usr/bin/env python3
"""Randomised differential run, general form (per-instance A, B, C, Q, R, bounds, x0, per-step
targets; I = 1, 2): LANE fp64 against the CPU oracle, bit for bit, iteration counts included."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle.bindings import Oracle, build_oracle
from trajectory_controller_amd import MpcSolver

build_oracle()
orc = Oracle()
rng = np.random.default_rng(int(os.environ.get("SEED", "11")))
sets = int(sys.argv[1]) if len(sys.argv) > 1 else 40
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
threads = int(os.environ.get("THREADS", "16"))
soa = lambda a: np.ascontiguousarray(np.asarray(a).reshape(a.shape[0], -1).T)
bad = 0
for s_i in range(sets):
    # every (I, H) pair in turn, so that no kernel is left to chance
    H = (5, 10, 20, 30, 40)[s_i % 5]
    I = 1 + (s_i // 5) % 2
    m = n if H <= 20 else n // 4
    A = np.tile(np.eye(2).reshape(1, 4), (m, 1)) + rng.normal(0, 0.08, (m, 4))
    B = rng.normal(0, 0.3, (m, 2 * I))
    Cc = rng.normal(0, 0.01, (m, 2))
    Q = 10 ** rng.uniform(-1, 1.5, (m, 2))
    if rng.random() < 0.3: Q[:, 1] = 0.0                       # dlib's own test has a zero weight
    R = 10 ** rng.uniform(-3, 1, (m, I))
    kind = rng.integers(0, 3)
    a = rng.uniform(0.05, 0.6, (m, I)); b = rng.uniform(0.05, 0.6, (m, I))
    if kind == 0:   lo, hi = -a, b
    elif kind == 1: lo, hi = np.zeros((m, I)), b                # start point on the lower bound
    else:           lo, hi = a, a + b                           # start point outside the bounds
    x0 = rng.normal(0, 0.2, (m, 2))
    targets = rng.normal(0, 0.3, (m, 1, 2)) + np.cumsum(rng.normal(0, 0.03, (m, H, 2)), axis=1)
    eps = float(10 ** rng.uniform(-4, -1.5)); cap = int(rng.choice([10000, 10000, 400, 51, 50]))
    # every other round of (I, H) pairs warm-starts from random controls: that is the state-returning
    # (unfused) projected-gradient kernel, and the whole solved control sequence is compared as well
    warm = (s_i // 10) % 2 == 1
    cin = rng.uniform(-0.7, 0.7, (m, H, I)) if warm else None
    u0, cout, it = orc.solve_general(I, H, A, B, Cc, Q, R, lo, hi, x0, targets, controls_in=cin, eps=eps,
                                     max_iter=cap, nthreads=threads)
    controls = soa(cin) if warm else None
    with MpcSolver(horizon=H, algo="lane", eps=eps, max_iter=cap) as s:
        gu0, git = s.solve_batch_general(soa(A), soa(B), soa(Cc), soa(Q), soa(R), soa(lo), soa(hi), soa(x0),
                                         soa(targets), controls=controls, inputs=I, want_iters=True)
    g = np.ascontiguousarray(gu0.T)
    diff = np.any(g.view(np.uint64) != np.ascontiguousarray(u0).view(np.uint64), axis=1) | (git != it)
    if warm:
        gc = np.ascontiguousarray(controls.T.reshape(m, H, I))
        diff |= np.any((gc.view(np.uint64) != np.ascontiguousarray(cout).view(np.uint64)).reshape(m, -1), axis=1)
    mism = int(np.sum(diff))
    bad += mism
    print(f"set {s_i:3d} I={I} H={H:2d} n={m:5d} {'warm' if warm else 'cold'} bounds kind {kind} eps {eps:.1e} cap {cap:5d} "
          f"mean iters {it.mean():7.1f} max {it.max():5d}: mismatching instances {mism}", flush=True)
print(f"{sets} parameter sets, total mismatches {bad}")
sys.exit(1 if bad else 0)
