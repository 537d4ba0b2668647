"""Synthetic trajectory-follower inputs.

A self-contained PRNG (splitmix64 -> (x >> 11) * 2^-53) so inputs are reproducible on any box
without libstdc++/numpy distributions: seed 0x5EED0000 + H; per instance three uniforms in the
order (v, dy, dphi):

    v    ~ U(0.1, 4.0)  m/s   speed after the velocity lookup (src/trajectory_point_follower.cpp:323)
    dy   ~ U(-0.5, 0.5) m     y_soll  (src/trajectory_point_follower.cpp:85)
    dphi ~ U(-0.6, 0.6) rad   phi_soll (src/trajectory_point_follower.cpp:84)
"""
from __future__ import annotations

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)
_GAMMA = np.uint64(0x9E3779B97F4A7C15)


def splitmix64_uniform(seed: int, count: int, offset: int = 0) -> np.ndarray:
    """`count` doubles in [0,1): element k is splitmix64 output number offset+k of stream `seed`."""
    with np.errstate(over="ignore"):
        k = np.arange(offset + 1, offset + count + 1, dtype=np.uint64)
        z = np.uint64(seed & 0xFFFFFFFFFFFFFFFF) + k * _GAMMA
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def compact_inputs(H: int, n: int, first: int = 0, seed: int | None = None):
    """(v, dy, dphi) for instances first .. first+n-1 of the horizon-H stream (fp64 arrays)."""
    seed = (0x5EED0000 + H) if seed is None else seed
    u = splitmix64_uniform(seed, 3 * n, offset=3 * first).reshape(n, 3)
    v = 0.1 + 3.9 * u[:, 0]
    dy = -0.5 + 1.0 * u[:, 1]
    dphi = -0.6 + 1.2 * u[:, 2]
    return np.ascontiguousarray(v), np.ascontiguousarray(dy), np.ascontiguousarray(dphi)


def general_inputs(H: int, n: int, I: int = 2, first: int = 0, seed: int | None = None,
                   T: float = 0.1, l: float = 0.21):
    """General-form instances: the compact model per instance plus per-step targets
    (compact target + smooth ramp), x0 ~ U(-0.1,0.1)^2 and a small constant disturbance C.
    Returns a dict of AoS fp64 arrays: A[n,4] B[n,2I] C[n,2] Q[n,2] R[n,I] lo[n,I] hi[n,I]
    x0[n,2] targets[n,H,2]."""
    seed = (0x5EED0000 + 0x1000 + H) if seed is None else seed
    v, dy, dphi = compact_inputs(H, n, first=first, seed=seed)
    e = splitmix64_uniform(seed ^ 0xA5A5A5A5, 6 * n, offset=6 * first).reshape(n, 6)
    A = np.stack([np.ones(n), T * v, np.zeros(n), np.ones(n)], axis=1)
    if I == 2:
        B = np.stack([np.zeros(n), T * v, T * v / l, -T * v / l], axis=1)
        R = np.tile(np.array([0.0005, 10.0]), (n, 1))
    else:
        B = np.stack([T * v, T * v / l], axis=1)
        R = np.tile(np.array([0.05]), (n, 1))
    Q = np.tile(np.array([20.0, 7.0]), (n, 1))
    amax = 22.0 * np.pi / 180.0
    lo = np.full((n, I), -amax)
    hi = np.full((n, I), amax)
    x0 = -0.1 + 0.2 * e[:, 0:2]
    Cc = (-0.5 + e[:, 2:4]) * 0.004
    ramp = np.linspace(0.0, 1.0, H)[None, :, None]
    slope = (-0.5 + e[:, 4:6])[:, None, :] * np.array([0.4, 0.5])[None, None, :]
    targets = np.stack([dy, dphi], axis=1)[:, None, :] + ramp * slope
    return dict(A=A, B=B, C=Cc, Q=Q, R=R, lo=lo, hi=hi, x0=x0,
                targets=np.ascontiguousarray(targets))
