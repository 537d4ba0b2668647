#!/usr/bin/env python3
"""Randomised differential run on the GPU box: the fp32 GROUP kernels (G lanes per instance, built for two wavefronts per
SIMD, two builds of the stop test: csrc/mpc_group.h, group_pg_kernel<..., MOVED>).

fp32 has no reference to be held to (dlib is fp64-only), so the statements are internal ones, each bit for bit:
  a) the two stop-test builds agree: a batch, and the same batch with ONE instance appended that fails the rounding
     screen (ub::moved_stop_ok: its speed is 50 m/s) and so sends the whole batch to the mask build, give every shared
     instance the same outputs and iteration counts;
  b) one and two wavefronts per SIMD agree (tpc_mpc_x_set_group_share pins the grid);
and, as a tolerance statement, the agreement with the fp32 LANE_FMA kernels (the same arithmetic in another
association; those are held bit for bit to their CPU model by fuzz_lane_fma.py): fraction of equal iteration counts and
the error distribution, reported per set.
Random weights, bounds (equal pairs -- fp32 keeps dlib's coordinates and has one build for them --, tight boxes), step
size, wheelbase, eps, iteration caps, speeds scaled into and out of the screen.  One line per parameter set; exits
non-zero if a) or b) fails anywhere.
    python tests/extended/fuzz_group_f32.py [sets] [n]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from trajectory_controller_amd import MpcSolver, capi
from trajectory_controller_amd.synth import compact_inputs

rng = np.random.default_rng(int(os.environ.get("SEED", "20261007")))
sets = int(sys.argv[1]) if len(sys.argv) > 1 else 60
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2500
BUILT = {10: (2, 4), 20: (2, 4, 8), 30: (2, 4, 8), 40: (2, 4, 8)}
cus = torch.cuda.get_device_properties(0).multi_processor_count
bad = total = 0
for s_i in range(sets):
    H = (10, 20, 20, 30, 40)[s_i % 5]
    G = BUILT[H][(s_i // 5) % len(BUILT[H])]
    m = n if H <= 20 else n // 4
    w = (10 ** rng.uniform(-1, 2), 10 ** rng.uniform(-1, 2), 10 ** rng.uniform(-4, 1), 10 ** rng.uniform(-2, 1.5))
    kind = int(rng.integers(0, 3))
    if kind == 0:   lo, hi = (-0.384, -0.384), (0.384, 0.384)
    elif kind == 1: a = float(rng.uniform(0.02, 0.6)); lo, hi = (-a, -a), (a, a)
    else:           a = float(rng.uniform(1e-3, 2e-2)); lo, hi = (-a, -a), (a, a)
    T = float(rng.uniform(0.02, 0.3)); l = float(rng.uniform(0.1, 0.5))
    eps = float(10 ** rng.uniform(-3, -1)); cap = int(rng.choice([10000, 10000, 3000, 300, 77, 51, 50]))
    smo = int(rng.choice([50, 50, 50, 0, 7, 120]))
    v, dy, dphi = (x.astype(np.float32) for x in compact_inputs(H, m, first=int(rng.integers(0, 1 << 30))))
    v = (v * np.float32(rng.choice([0.25, 0.5, 0.75, 1.0, 2.0]))).astype(np.float32)   # (2.0: beyond the screen by itself at N >= 20)
    scale = np.float32(rng.choice([1.0, 1.0, 0.2, 3.0]))
    dy, dphi = dy * scale, dphi * scale
    v2, dy2, dphi2 = (np.concatenate([x, x[:1]]) for x in (v, dy, dphi))
    v2[-1] = np.float32(50.0)
    skw = dict(weight_y=w[0], weight_phi=w[1], weight_steering_front=w[2], weight_steering_rear=w[3], lower=lo, upper=hi,
               step_size=T, wheelbase=l, eps=eps, max_iter=cap, smo_iters=smo, dtype="f32")
    with MpcSolver(horizon=H, algo="group", **skw) as s:
        s.set_option(capi.OPT_GROUP_LANES, G)
        f, r, it = s.solve_batch_compact(v, dy, dphi, want_iters=True)
        fm, rm, itm = s.solve_batch_compact(v2, dy2, dphi2, want_iters=True)
        grids = []
        for per_simd in (1, 2):
            s._check(s._lib.tpc_mpc_x_set_group_share(s._h, per_simd * 4 * cus, 0))
            grids.append(s.solve_batch_compact(v, dy, dphi, want_iters=True))
    with MpcSolver(horizon=H, algo="lane_fma", **skw) as s:
        lf, lr, lit = s.solve_batch_compact(v, dy, dphi, want_iters=True)
    builds_ok = np.array_equal(it, itm[:m]) and np.array_equal(f, fm[:m]) and np.array_equal(r, rm[:m])
    grids_ok = all(np.array_equal(it, gi) and np.array_equal(f, gf) and np.array_equal(r, gr) for gf, gr, gi in grids)
    err = np.maximum(np.abs(f.astype(np.float64) - lf), np.abs(r.astype(np.float64) - lr))
    bad += (not builds_ok) + (not grids_ok)
    total += m
    print(f"set {s_i:3d} H={H:2d} G={G} n={m:5d} bounds kind {kind} eps {eps:.1e} cap {cap:5d} smo {smo:3d}: stop-test builds "
          f"{'agree' if builds_ok else 'DIFFER'}, one / two wavefronts per SIMD {'agree' if grids_ok else 'DIFFER'}; against LANE_FMA fp32: "
          f"equal iteration counts {np.mean(it == lit):.4f}, |du| median {np.median(err):.1e} p99 {np.quantile(err, 0.99):.1e} max {err.max():.1e}; "
          f"capped {int((it >= cap).sum())}", flush=True)
print(f"{sets} parameter sets, {total} instances: {bad} disagreements between the stop-test builds or the grid sizes")
sys.exit(1 if bad else 0)
