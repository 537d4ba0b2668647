"""oracle/bindings.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

ctypes bindings for the two CPU checkers:

* ``Oracle``  -> oracle/liboracle_mpc.so, the plain-C restatement (oracle/mpc_oracle.c) of
  dlib::mpc (reference: dlib_files/dlib/control/mpc.h:51-347).
* ``DlibRef`` -> oracle/_ref/libdlib_mpc_ref.so, the REAL dlib::mpc compiled from the reference's
  own headers (oracle/ref_dlib_harness.cpp); exists only where `make -C oracle ref` has run in a
  container holding /root/reference, and travels to the GPU box as a prebuilt .so.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product package (trajectory_controller_amd) never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(_HERE, "liboracle_mpc.so")
ORACLE_F32_SO = os.path.join(_HERE, "liboracle_mpc_f32.so")   # the same source typed float
REF_SO = os.path.join(_HERE, "_ref", "libdlib_mpc_ref.so")

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


def _d(a):
    return None if a is None else a.ctypes.data_as(_dp)


def _i(a):
    return None if a is None else a.ctypes.data_as(_ip)


def _f64(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None:
        a = a.reshape(shape)
    return a


def build_oracle(force: bool = False) -> str:
    """Compile the C restatement (gcc) if missing or stale."""
    src = os.path.join(_HERE, "mpc_oracle.c")
    hdr = os.path.join(_HERE, "mpc_oracle.h")
    stale = any((not os.path.exists(so)) or any(os.path.getmtime(p) > os.path.getmtime(so) for p in (src, hdr))
                for so in (ORACLE_SO, ORACLE_F32_SO))
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle_mpc.so", "liboracle_mpc_f32.so"],
                              stdout=subprocess.DEVNULL)
    return ORACLE_SO


def build_ref(reference: str = "/root/reference") -> str | None:
    """Compile real dlib::mpc from the reference tree when it is present; else keep a prebuilt."""
    if os.path.isdir(os.path.join(reference, "dlib_files")):
        src = os.path.join(_HERE, "ref_dlib_harness.cpp")
        if (not os.path.exists(REF_SO)) or os.path.getmtime(src) > os.path.getmtime(REF_SO):
            subprocess.check_call(["make", "-C", _HERE, "-B", "ref", f"REFERENCE={reference}"],
                                  stdout=subprocess.DEVNULL)
    return REF_SO if os.path.exists(REF_SO) else None


DEFAULT_WEIGHTS = (20.0, 7.0, 0.0005, 10.0)       # src/trajectory_point_follower.cpp:92-95
ALPHA_MAX = 22.0 * np.pi / 180.0                   # src/trajectory_point_follower.cpp:16
DEFAULT_T = 0.1                                    # src/trajectory_point_follower.cpp:96
DEFAULT_L = 0.21                                   # include/trajectory_point_follower.h:47


class Oracle:
    """The C restatement.  dtype "f64" (default): the checker pinned to real dlib.  dtype "f32":
    the same source compiled with every value typed float -- what "the same operation sequence in
    fp32" means for the fp32 kernels (dlib itself is fp64-only, so that build is unpinned)."""

    def __init__(self, dtype: str = "f64"):
        build_oracle()
        self.dtype = dtype
        self.np = {"f64": np.float64, "f32": np.float32}[dtype]
        self.lib = C.CDLL(ORACLE_SO if dtype == "f64" else ORACLE_F32_SO)
        rp = C.POINTER(C.c_double if dtype == "f64" else C.c_float)
        self._rp = rp
        L = self.lib
        L.mpc_oracle_solve_compact.restype = C.c_int
        L.mpc_oracle_solve_compact.argtypes = [
            C.c_int, C.c_long, C.c_int, rp, rp, rp, rp, C.c_double, C.c_double, rp, rp,
            C.c_double, C.c_ulong, C.c_ulong, rp, rp, _ip]
        L.mpc_oracle_solve_general_state.restype = C.c_int
        L.mpc_oracle_solve_general_state.argtypes = [
            C.c_int, C.c_int, C.c_long, C.c_int, rp, rp, rp, rp, rp, rp, rp, rp, rp, rp, rp,
            C.c_double, C.c_ulong, C.c_ulong, rp, rp, rp, _ip]
        L.mpc_oracle_rollout.restype = C.c_int
        L.mpc_oracle_rollout.argtypes = [
            C.c_int, C.c_int, C.c_int, rp, rp, rp, rp, rp, rp, rp, rp, rp, rp,
            C.c_double, C.c_ulong, C.c_ulong, rp, rp, _ip]

    def _a(self, a, shape=None):
        a = np.ascontiguousarray(a, dtype=self.np)
        return a if shape is None else a.reshape(shape)

    def _p(self, a):
        return None if a is None else a.ctypes.data_as(self._rp)

    def solve_compact(self, H, v, dy, dphi, weights=DEFAULT_WEIGHTS, T=DEFAULT_T, l=DEFAULT_L,
                      lo=(-ALPHA_MAX, -ALPHA_MAX), hi=(ALPHA_MAX, ALPHA_MAX), eps=0.01,
                      max_iter=10000, smo_iters=50, nthreads=1):
        _a, _p = self._a, self._p
        v, dy, dphi = _a(v), _a(dy), _a(dphi)
        n = v.shape[0]
        w, lo, hi = _a(weights), _a(lo), _a(hi)
        front, rear = np.empty(n, dtype=self.np), np.empty(n, dtype=self.np)
        iters = np.empty(n, dtype=np.int32)
        rc = self.lib.mpc_oracle_solve_compact(H, n, nthreads, _p(v), _p(dy), _p(dphi), _p(w), T,
                                               l, _p(lo), _p(hi), eps, max_iter, smo_iters,
                                               _p(front), _p(rear), _i(iters))
        if rc != 0:
            raise ValueError(f"oracle: unsupported H={H}")
        return front, rear, iters

    def solve_general(self, I, H, A, B, Cc, Q, R, lo, hi, x0, targets, controls_in=None,
                      eps=0.01, max_iter=10000, smo_iters=50, nthreads=1, v_in=None, want_v=False):
        """Returns (u0[n,I], controls[n,H,I], iters[n]) and, with want_v, dlib's v[n,H,I] too."""
        _a, _p = self._a, self._p
        A = _a(A).reshape(-1, 4)
        n = A.shape[0]
        B, Cc, Q, R = _a(B, (n, 2 * I)), _a(Cc, (n, 2)), _a(Q, (n, 2)), _a(R, (n, I))
        lo, hi, x0 = _a(lo, (n, I)), _a(hi, (n, I)), _a(x0, (n, 2))
        targets = _a(targets, (n, H, 2))
        cin = None if controls_in is None else _a(controls_in, (n, H, I))
        vin = None if v_in is None else _a(v_in, (n, H, I))
        u0 = np.empty((n, I), dtype=self.np)
        cout = np.empty((n, H, I), dtype=self.np)
        vout = np.empty((n, H, I), dtype=self.np) if want_v else None
        iters = np.empty(n, dtype=np.int32)
        rc = self.lib.mpc_oracle_solve_general_state(
            I, H, n, nthreads, _p(A), _p(B), _p(Cc), _p(Q), _p(R), _p(lo), _p(hi), _p(x0),
            _p(targets), _p(cin), _p(vin), eps, max_iter, smo_iters, _p(u0), _p(cout), _p(vout),
            _i(iters))
        if rc != 0:
            raise ValueError(f"oracle: unsupported I={I} H={H}")
        return (u0, cout, iters, vout) if want_v else (u0, cout, iters)

    def rollout(self, I, H, steps, A, B, Cc, Q, R, lo, hi, x0, targets0, new_last_targets=None,
                eps=0.01, max_iter=10000, smo_iters=50):
        _a, _p = self._a, self._p
        A, B, Cc, Q, R = _a(A), _a(B), _a(Cc), _a(Q), _a(R)
        lo, hi, x0, targets0 = _a(lo), _a(hi), _a(x0), _a(targets0, (H, 2))
        nlt = None if new_last_targets is None else _a(new_last_targets, (steps, 2))
        controls = np.empty((steps, I), dtype=self.np)
        states = np.empty((steps, 2), dtype=self.np)
        iters = np.empty(steps, dtype=np.int32)
        rc = self.lib.mpc_oracle_rollout(I, H, steps, _p(A), _p(B), _p(Cc), _p(Q), _p(R), _p(lo),
                                         _p(hi), _p(x0), _p(targets0), _p(nlt), eps, max_iter,
                                         smo_iters, _p(controls), _p(states), _i(iters))
        if rc != 0:
            raise ValueError(f"oracle: unsupported I={I} H={H}")
        return controls, states, iters


class DlibRef:
    """Real dlib::mpc.  Supported horizons: 4, 5, 10, 20, 30, 40 (template instantiations)."""
    HORIZONS = (4, 5, 10, 20, 30, 40)

    def __init__(self, path: str | None = None):
        path = path or build_ref()
        if path is None or not os.path.exists(path):
            raise FileNotFoundError("oracle/_ref/libdlib_mpc_ref.so not built (no reference tree)")
        self.lib = C.CDLL(path)
        L = self.lib
        L.dlibref_solve_compact.restype = C.c_int
        L.dlibref_solve_compact.argtypes = [
            C.c_int, C.c_long, C.c_int, _dp, _dp, _dp, _dp, C.c_double, C.c_double, _dp, _dp,
            C.c_double, C.c_ulong, _dp, _dp]
        L.dlibref_solve_general.restype = C.c_int
        L.dlibref_solve_general.argtypes = [
            C.c_int, C.c_int, C.c_long, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp,
            C.c_double, C.c_ulong, _dp]
        L.dlibref_rollout.restype = C.c_int
        L.dlibref_rollout.argtypes = [
            C.c_int, C.c_int, C.c_int, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp,
            C.c_double, C.c_ulong, _dp, _dp]

    @staticmethod
    def available() -> bool:
        return build_ref() is not None

    def solve_compact(self, H, v, dy, dphi, weights=DEFAULT_WEIGHTS, T=DEFAULT_T, l=DEFAULT_L,
                      lo=(-ALPHA_MAX, -ALPHA_MAX), hi=(ALPHA_MAX, ALPHA_MAX), eps=0.01,
                      max_iter=10000, nthreads=1):
        v, dy, dphi = _f64(v), _f64(dy), _f64(dphi)
        n = v.shape[0]
        w, lo, hi = _f64(weights), _f64(lo), _f64(hi)
        front, rear = np.empty(n), np.empty(n)
        rc = self.lib.dlibref_solve_compact(H, n, nthreads, _d(v), _d(dy), _d(dphi), _d(w), T, l,
                                            _d(lo), _d(hi), eps, max_iter, _d(front), _d(rear))
        if rc != 0:
            raise ValueError(f"dlibref: unsupported H={H}")
        return front, rear

    def solve_general(self, I, H, A, B, Cc, Q, R, lo, hi, x0, targets, eps=0.01, max_iter=10000):
        A = _f64(A).reshape(-1, 4)
        n = A.shape[0]
        B, Cc, Q, R = _f64(B, (n, 2 * I)), _f64(Cc, (n, 2)), _f64(Q, (n, 2)), _f64(R, (n, I))
        lo, hi, x0 = _f64(lo, (n, I)), _f64(hi, (n, I)), _f64(x0, (n, 2))
        targets = _f64(targets, (n, H, 2))
        u0 = np.empty((n, I))
        rc = self.lib.dlibref_solve_general(I, H, n, _d(A), _d(B), _d(Cc), _d(Q), _d(R), _d(lo),
                                            _d(hi), _d(x0), _d(targets), eps, max_iter, _d(u0))
        if rc != 0:
            raise ValueError(f"dlibref: unsupported I={I} H={H}")
        return u0

    def rollout(self, I, H, steps, A, B, Cc, Q, R, lo, hi, x0, targets0, new_last_targets=None,
                eps=0.01, max_iter=10000):
        A, B, Cc, Q, R = _f64(A), _f64(B), _f64(Cc), _f64(Q), _f64(R)
        lo, hi, x0, targets0 = _f64(lo), _f64(hi), _f64(x0), _f64(targets0, (H, 2))
        nlt = None if new_last_targets is None else _f64(new_last_targets, (steps, 2))
        controls = np.empty((steps, I))
        states = np.empty((steps, 2))
        rc = self.lib.dlibref_rollout(I, H, steps, _d(A), _d(B), _d(Cc), _d(Q), _d(R), _d(lo),
                                      _d(hi), _d(x0), _d(targets0), _d(nlt), eps, max_iter,
                                      _d(controls), _d(states))
        if rc != 0:
            raise ValueError(f"dlibref: unsupported I={I} H={H}")
        return controls, states
