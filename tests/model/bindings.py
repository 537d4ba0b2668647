"""tests/model/bindings.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

ctypes binding of the CPU model of the LANE_FMA kernel family (tests/model/ub_model.cpp): the same
IEEE operations as the gfx950 kernels of that family, so GPU results are compared with it bit for bit.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
MODEL_SO = os.path.join(_HERE, "libub_model.so")
_ROOT = os.path.dirname(os.path.dirname(_HERE))
_SRCS = [os.path.join(_HERE, "ub_model.cpp"),
         os.path.join(_ROOT, "trajectory_controller_amd", "csrc", "mpc_ub_model.h")]

ALPHA_MAX = 22.0 * np.pi / 180.0
DEFAULT_WEIGHTS = (20.0, 7.0, 0.0005, 10.0)


def build_model(force: bool = False) -> str:
    stale = (not os.path.exists(MODEL_SO)) or any(
        os.path.exists(p) and os.path.getmtime(p) > os.path.getmtime(MODEL_SO) for p in _SRCS)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-B", "libub_model.so"], stdout=subprocess.DEVNULL)
    return MODEL_SO


class UbModel:
    def __init__(self, dtype: str = "f64"):
        build_model()
        self.dtype = dtype
        self.np = {"f64": np.float64, "f32": np.float32}[dtype]
        self.lib = C.CDLL(MODEL_SO)
        rp = C.POINTER(C.c_double if dtype == "f64" else C.c_float)
        self._rp = rp
        self.fn = getattr(self.lib, f"ub_model_solve_compact_{dtype}")
        self.fn.restype = C.c_int
        self.fn.argtypes = [C.c_int, C.c_long, C.c_int, rp, rp, rp, rp, C.c_double, C.c_double, rp, rp,
                            C.c_double, C.c_ulong, C.c_ulong, C.c_int, rp, rp, C.POINTER(C.c_int),
                            C.POINTER(C.c_uint)]
        self.gen = getattr(self.lib, f"ub_model_solve_general_{dtype}")
        self.gen.restype = C.c_int
        self.gen.argtypes = [C.c_int, C.c_int, C.c_long, C.c_int, rp, rp, rp, rp, rp, rp, rp, rp, rp,
                             C.c_double, C.c_ulong, C.c_ulong, C.c_int, rp, C.POINTER(C.c_int), C.POINTER(C.c_uint)]

    def solve_compact(self, H, v, dy, dphi, weights=DEFAULT_WEIGHTS, T=0.1, l=0.21,
                      lo=(-ALPHA_MAX, -ALPHA_MAX), hi=(ALPHA_MAX, ALPHA_MAX), eps=0.01, max_iter=10000,
                      smo_iters=50, nthreads=1, fast_stop=True):
        """fast_stop: True / False = a screened build (fp32: which of the two, decided batch-wide by the second screen like the
        kernels do) / the exact stop-test build; None = decided by the kernels' own screens."""
        a = lambda z: np.ascontiguousarray(z, dtype=self.np)
        p = lambda z: z.ctypes.data_as(self._rp)
        v, dy, dphi, w, lo, hi = a(v), a(dy), a(dphi), a(weights), a(lo), a(hi)
        n = v.shape[0]
        front, rear = np.empty(n, dtype=self.np), np.empty(n, dtype=self.np)
        iters = np.empty(n, dtype=np.int32)
        flags = C.c_uint(0)
        rc = self.fn(H, n, nthreads, p(v), p(dy), p(dphi), p(w), T, l, p(lo), p(hi), eps, max_iter, smo_iters,
                     -1 if fast_stop is None else int(fast_stop), p(front), p(rear), iters.ctypes.data_as(C.POINTER(C.c_int)),
                     C.byref(flags))
        if rc != 0:
            raise ValueError(f"ub model: unsupported H={H}")
        return front, rear, iters, flags.value

    def solve_general(self, I, H, A, B, Cc, Q, R, lo, hi, x0, targets, eps=0.01, max_iter=10000, smo_iters=50,
                      nthreads=1, fast_stop=None):
        """General form, cold start (AoS arrays like oracle.solve_general): returns (u0[n,I], iters[n], flags)."""
        a = lambda z, shape: np.ascontiguousarray(z, dtype=self.np).reshape(shape)
        p = lambda z: z.ctypes.data_as(self._rp)
        A = a(A, (-1, 4)); n = A.shape[0]
        B, Cc, Q, R = a(B, (n, 2 * I)), a(Cc, (n, 2)), a(Q, (n, 2)), a(R, (n, I))
        lo, hi, x0, targets = a(lo, (n, I)), a(hi, (n, I)), a(x0, (n, 2)), a(targets, (n, H, 2))
        u0 = np.empty((n, I), dtype=self.np)
        iters = np.empty(n, dtype=np.int32)
        flags = C.c_uint(0)
        rc = self.gen(I, H, n, nthreads, p(A), p(B), p(Cc), p(Q), p(R), p(lo), p(hi), p(x0), p(targets), eps, max_iter,
                      smo_iters, -1 if fast_stop is None else (1 if fast_stop else 0), p(u0),
                      iters.ctypes.data_as(C.POINTER(C.c_int)), C.byref(flags))
        if rc != 0:
            raise ValueError(f"ub model: unsupported I={I} H={H}")
        return u0, iters, flags.value
