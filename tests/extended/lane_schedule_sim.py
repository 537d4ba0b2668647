#!/usr/bin/env python3
"""CPU model of the LANE projected-gradient kernel's scheduler: persistent
waves of 64 lanes pull instances from one queue; a wave-iteration costs the same whatever the number
of active lanes; free lanes are refilled when K of them wait.  Iteration counts come from the CPU
oracle, so this is a checker-side tool: it reproduces the measured wave-iteration totals to 0.1 %
and was used to evaluate queue orders (lambda, the floor rule, learned keys, exact counts) and
refill batch sizes before touching the kernel.

    python tests/extended/lane_schedule_sim.py [H] [n]        (defaults 20, 65536; 4 instances per lane)
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle.bindings import Oracle, build_oracle, DEFAULT_WEIGHTS, DEFAULT_T, DEFAULT_L
from trajectory_controller_amd.synth import compact_inputs


def lambda_compact(H, v, w=DEFAULT_WEIGHTS, T=DEFAULT_T, l=DEFAULT_L):
    """dlib's trace bound (mpc.h:116-123) for the compact model, vectorised."""
    q0, q1, r0, r1 = w
    a, c = T * v, T * v / l
    lam = H * (r0 + r1) * np.ones(len(v))
    t00, t01, t10, t11 = np.full(len(v), q0), np.zeros(len(v)), np.zeros(len(v)), np.full(len(v), q1)
    for _ in range(H):
        lam += c * t11 * c + a * (t00 * a + t01 * (-c)) + (-c) * (t10 * a + t11 * (-c))
        u10, u11 = a * t00 + t10, a * t01 + t11
        t00, t01, t10, t11 = t00 + q0, t00 * a + t01, u10, u10 * a + u11 + q1
    return lam


def simulate(pg, order, K=3, lanes_per_instance=4):
    """Returns (wave-iterations per wave, refill passes per wave)."""
    n = len(pg)
    waves = n // lanes_per_instance // 64
    q = pg[order]
    rem = np.zeros((waves, 64), dtype=np.int64)
    rem.flat[:] = q[:waves * 64]
    qi = waves * 64
    wave_iters = np.zeros(waves, dtype=np.int64)
    refills = 0
    while True:
        active = rem > 0
        busy = active.any(axis=1)
        if not busy.any() and qi >= n:
            break
        step = np.where(active, rem, 1 << 60).min()
        rem = np.where(active, rem - step, rem)
        wave_iters += np.where(busy, step, 0)
        free = rem <= 0
        for w in np.nonzero(free.any(axis=1))[0]:
            idx = np.nonzero(free[w])[0]
            if qi >= n or (len(idx) < K and (rem[w] > 0).any()):
                continue
            take = min(len(idx), n - qi)
            rem[w, idx[:take]] = q[qi:qi + take]
            qi += take
            refills += 1
    return wave_iters, refills / waves


def main():
    H = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
    build_oracle()
    v, dy, dphi = compact_inputs(H, n)
    _, _, it = Oracle().solve_compact(H, v, dy, dphi, nthreads=os.cpu_count() or 8)
    pg = np.maximum(it - 50, 0).astype(np.int64)          # projected-gradient iterations
    lam = lambda_compact(H, v)
    floor = H * (DEFAULT_WEIGHTS[2] + DEFAULT_WEIGHTS[3])
    by_lambda = np.argsort(-lam, kind="stable")
    low = lam < 1.5 * floor
    orders = {
        "random": np.random.default_rng(0).permutation(n),
        "lambda descending": by_lambda,
        "lambda + floor rule (the kernel's order)": np.concatenate([by_lambda[low[by_lambda]], by_lambda[~low[by_lambda]]]),
        "exact iteration counts": np.argsort(-pg, kind="stable"),
    }
    print(f"H={H} n={n}: mean PG iterations {pg.mean():.0f}, max {pg.max()}, perfectly balanced wave-iterations {pg.sum() / (n // 4):.0f}")
    for K in (1, 3):
        for name, order in orders.items():
            wi, rf = simulate(pg, order, K=K)
            print(f"  K={K} {name:42s} slowest wave {wi.max():6d}  mean {wi.mean():8.1f}  "
                  f"lane utilisation {pg.sum() / (64.0 * wi.sum()):.4f}  refill passes/wave {rf:6.1f}")


if __name__ == "__main__":
    main()
