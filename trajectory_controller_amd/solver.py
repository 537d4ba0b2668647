"""Host-side handle over the C ABI: `MpcSolver`.

Mirrors the reference's operator surface for this path -- dlib::mpc's constructor knobs
(epsilon, max_iterations; mpc.h:187-214), `mpcControllerTobi(v, delta_y, delta_phi)`
(src/trajectory_point_follower.cpp:301-389) -- for one instance or a batch.  Arrays may be numpy
(host memory: the library stages them) or torch tensors on the GPU (device pointers are handed to
the library as they are, launches go to torch's current stream).  torch is used for device memory
and streams only; all arithmetic is in the HIP kernels behind libtpc_mpc.so.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np

from . import capi

_NP = {capi.F64: np.float64, capi.F32: np.float32}


def _is_torch(x) -> bool:
    return type(x).__module__.startswith("torch")


class MpcSolver:
    """One tpc_mpc_handle bound to a HIP device (device=None: host-only, TPC_MPC_DEVICE_NONE).  Not thread-safe (like the handle)."""

    def __init__(self, horizon: int = 20, device: int = 0, dtype: str = "f64", algo: str = "auto",
                 **params):
        self._lib = capi.load_library()
        self._h = C.c_void_p()
        device = capi.DEVICE_NONE if device is None else int(device)   # None: a host-only handle (solve_one only, no GPU touched)
        rc = self._lib.tpc_mpc_create(device, C.byref(self._h))
        if rc != capi.OK:
            raise capi.TpcMpcError(rc, self._lib.tpc_mpc_last_error(None).decode())
        self.device = device
        self.dtype = {"f64": capi.F64, "f32": capi.F32}[dtype]
        self.algo = {"auto": capi.ALGO_AUTO, "wave": capi.ALGO_WAVE, "lane": capi.ALGO_LANE,
                     "lane_fma": capi.ALGO_LANE_FMA, "group": capi.ALGO_GROUP}[algo]
        self.params = capi.default_params(horizon, self.dtype, self.algo, **params)
        self.last_flags = 0
        self.rank, self.world = 0, 1     # a handle without a communicator is a world of one

    # -- lifetime ---------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._lib.tpc_mpc_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    @property
    def horizon(self) -> int:
        return self.params.horizon

    def _check(self, rc: int):
        if rc != capi.OK:
            raise capi.TpcMpcError(rc, self._lib.tpc_mpc_last_error(self._h).decode())

    def _params(self, **over) -> capi.Params:
        if not over:
            return self.params
        p = capi.Params.from_buffer_copy(self.params)
        for k, val in over.items():
            if k in ("lower", "upper"):
                getattr(p, k)[0], getattr(p, k)[1] = float(val[0]), float(val[1])
            else:
                setattr(p, k, val)
        return p

    # -- measurement ------------------------------------------------------------------------------
    def set_profiling(self, enable: bool = True):
        self._check(self._lib.tpc_mpc_set_profiling(self._h, int(enable)))

    def last_kernel_times(self):
        """(first_kernel_ms, second_kernel_ms, algo) of the last solve; waits for it."""
        a, b, algo = C.c_double(), C.c_double(), C.c_int()
        self._check(self._lib.tpc_mpc_last_kernel_times(self._h, C.byref(a), C.byref(b), C.byref(algo)))
        return a.value, b.value, algo.value

    def last_lane_stats(self):
        """(wave_iterations, refill_blocks) of the last LANE solve; synchronises the device."""
        a, b = C.c_uint64(), C.c_uint64()
        self._check(self._lib.tpc_mpc_last_lane_stats(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    # -- the call being replaced ------------------------------------------------------------------
    def mpc_controller_tobi(self, v: float, delta_y: float, delta_phi: float, **over):
        """mpcControllerTobi(v, delta_y, delta_phi, &front, &rear): returns (front, rear)."""
        f, r = C.c_double(), C.c_double()
        self._check(self._lib.tpc_mpc_solve_one(self._h, C.byref(self._params(**over)), float(v),
                                                float(delta_y), float(delta_phi), C.byref(f),
                                                C.byref(r)))
        return f.value, r.value

    solve_one = mpc_controller_tobi

    def last_solve_one_flags(self):
        """(flags, iterations) of the last solve_one (tpc_mpc_last_flags): capi.FLAG_NONFINITE when the speed
        or the target was NaN / Inf (the call returned dlib's untouched start point), capi.FLAG_MAX_ITER when
        max_iter cut the solve off."""
        f, it = C.c_uint32(), C.c_int32()
        self._check(self._lib.tpc_mpc_last_flags(self._h, C.byref(f), C.byref(it)))
        return f.value, it.value

    def set_option(self, option: int, value: int):
        """tpc_mpc_set_option: capi.OPT_WAVE_GROUP (0 auto, 1, 2, 4), capi.OPT_MAILBOX_HOST (0 / 1)."""
        self._check(self._lib.tpc_mpc_set_option(self._h, int(option), int(value)))

    def set_resident(self, idle_timeout_us: int = 20000):
        """solve_one's resident wavefront: idle timeout in microseconds, <= 0 turns it off
        (tpc_mpc_set_resident)."""
        self._check(self._lib.tpc_mpc_set_resident(self._h, int(idle_timeout_us)))

    @staticmethod
    def build_info() -> str:
        return capi.load_library().tpc_mpc_build_info().decode()

    # -- batches ----------------------------------------------------------------------------------
    def reserve(self, n: int, host: bool = False, **over):
        """Allocate the device scratch for compact batches of up to n instances now
        (tpc_mpc_reserve), so that no later solve allocates -- needed when several batches are kept
        in flight on different streams, because a device allocation synchronises all of them."""
        p = self._params(**over)
        self._check(self._lib.tpc_mpc_reserve(self._h, C.byref(p), int(n), capi.HOST if host else capi.DEVICE))

    def set_work_hint(self, hint):
        """Queue-order hint for the next batch solve of the same size (experimental: tpc_mpc_x_set_work_hint,
        csrc/tpc_mpc_experimental.h):
        per-instance iteration-count estimates, typically the `iters` of the previous cycle.
        int32 numpy array (copied) or CUDA tensor (read by the next solve; keep it alive until
        then).  None clears.  Never changes a result, only the order lanes pick instances up."""
        if hint is None:
            self._check(self._lib.tpc_mpc_x_set_work_hint(self._h, None, 0, capi.HOST))
            self._hint_ref = None
        elif _is_torch(hint):
            import torch
            if not (hint.is_cuda and hint.dtype == torch.int32 and hint.is_contiguous()):
                raise ValueError("a device hint must be a contiguous int32 CUDA tensor")
            self._hint_ref = hint
            self._check(self._lib.tpc_mpc_x_set_work_hint(self._h, hint.data_ptr(), hint.numel(), capi.DEVICE))
        else:
            h = np.ascontiguousarray(hint, dtype=np.int32)
            self._check(self._lib.tpc_mpc_x_set_work_hint(self._h, h.ctypes.data, h.shape[0], capi.HOST))

    def solve_batch_compact(self, v, delta_y, delta_phi, want_iters: bool = False,
                            want_flags: bool = True, out=None, **over):
        """n independent mpcControllerTobi calls.  Returns (front, rear[, iters])."""
        p = self._params(**over)
        if _is_torch(v):
            import torch
            tdt = torch.float64 if p.dtype == capi.F64 else torch.float32
            n = v.numel()
            for t in (v, delta_y, delta_phi):
                if not (t.is_cuda and t.dtype == tdt and t.is_contiguous() and t.numel() == n):
                    raise ValueError("device batch arrays must be contiguous CUDA tensors of the solver dtype")
            if out is None:
                front, rear = torch.empty_like(v), torch.empty_like(v)
            else:
                front, rear = out
            iters = torch.empty(n, dtype=torch.int32, device=v.device) if want_iters else None
            flags = C.c_uint32(0)
            stream = torch.cuda.current_stream(v.device).cuda_stream
            self._check(self._lib.tpc_mpc_solve_batch_compact(
                self._h, C.byref(p), n, v.data_ptr(), delta_y.data_ptr(), delta_phi.data_ptr(),
                front.data_ptr(), rear.data_ptr(), iters.data_ptr() if want_iters else None,
                C.byref(flags) if want_flags else None, capi.DEVICE, C.c_void_p(stream)))
            self.last_flags = flags.value
        else:
            dt = _NP[p.dtype]
            v, delta_y, delta_phi = (np.ascontiguousarray(a, dtype=dt) for a in (v, delta_y, delta_phi))
            n = v.shape[0]
            front, rear = np.empty(n, dtype=dt), np.empty(n, dtype=dt)
            iters = np.empty(n, dtype=np.int32) if want_iters else None
            flags = C.c_uint32(0)
            self._check(self._lib.tpc_mpc_solve_batch_compact(
                self._h, C.byref(p), n, v.ctypes.data, delta_y.ctypes.data, delta_phi.ctypes.data,
                front.ctypes.data, rear.ctypes.data, iters.ctypes.data if want_iters else None,
                C.byref(flags), capi.HOST, None))
            self.last_flags = flags.value
        return (front, rear, iters) if want_iters else (front, rear)

    def solve_batch_compact_mixed(self, horizons, v, delta_y, delta_phi, want_iters: bool = False, **over):
        """Mixed-horizon batch (BASELINE config 5): instance k is solved with horizon horizons[k]
        (tpc_mpc_solve_batch_compact_mixed: binned by horizon on the device, one launch sequence per
        bin, results back in the caller's order).  Arrays as in solve_batch_compact; `horizons` is an
        integer array/tensor of the same length."""
        p = self._params(**over)
        flags = C.c_uint32(0)
        if _is_torch(v):
            import torch
            tdt = torch.float64 if p.dtype == capi.F64 else torch.float32
            n = v.numel()
            for t in (v, delta_y, delta_phi):
                if not (t.is_cuda and t.dtype == tdt and t.is_contiguous() and t.numel() == n):
                    raise ValueError("device batch arrays must be contiguous CUDA tensors of the solver dtype")
            hz = torch.as_tensor(horizons, device=v.device).to(torch.int32).contiguous()
            if hz.numel() != n:
                raise ValueError("horizons must have one entry per instance")
            front, rear = torch.empty_like(v), torch.empty_like(v)
            iters = torch.empty(n, dtype=torch.int32, device=v.device) if want_iters else None
            stream = torch.cuda.current_stream(v.device).cuda_stream
            self._check(self._lib.tpc_mpc_solve_batch_compact_mixed(
                self._h, C.byref(p), n, hz.data_ptr(), v.data_ptr(), delta_y.data_ptr(), delta_phi.data_ptr(),
                front.data_ptr(), rear.data_ptr(), iters.data_ptr() if want_iters else None,
                C.byref(flags), capi.DEVICE, C.c_void_p(stream)))
        else:
            dt = _NP[p.dtype]
            v, delta_y, delta_phi = (np.ascontiguousarray(a, dtype=dt) for a in (v, delta_y, delta_phi))
            n = v.shape[0]
            hz = np.ascontiguousarray(horizons, dtype=np.int32)
            if hz.shape[0] != n:
                raise ValueError("horizons must have one entry per instance")
            front, rear = np.empty(n, dtype=dt), np.empty(n, dtype=dt)
            iters = np.empty(n, dtype=np.int32) if want_iters else None
            self._check(self._lib.tpc_mpc_solve_batch_compact_mixed(
                self._h, C.byref(p), n, hz.ctypes.data, v.ctypes.data, delta_y.ctypes.data, delta_phi.ctypes.data,
                front.ctypes.data, rear.ctypes.data, iters.ctypes.data if want_iters else None,
                C.byref(flags), capi.HOST, None))
        self.last_flags = flags.value
        return (front, rear, iters) if want_iters else (front, rear)

    # -- sharding over the GPUs of a node ----------------------------------------------------------
    @staticmethod
    def comm_unique_id() -> bytes:
        """128-byte communicator id (rank 0 makes it, the host hands it to every rank)."""
        lib = capi.load_library()
        buf = (C.c_char * capi.COMM_ID_BYTES)()
        rc = lib.tpc_mpc_comm_unique_id(buf, capi.COMM_ID_BYTES)
        if rc != capi.OK:
            raise capi.TpcMpcError(rc, lib.tpc_mpc_last_error(None).decode())
        return bytes(buf)

    def comm_test_mode(self, force_communicator: bool, force_ragged: bool = False):
        """Test hook (tpc_mpc_comm_test_mode): a real one-rank communicator / the ragged exchange on one GPU."""
        self._check(self._lib.tpc_mpc_comm_test_mode(self._h, int(force_communicator), int(force_ragged)))

    def comm_init(self, comm_id: bytes, rank: int, world: int):
        """Join the job's RCCL communicator as `rank` of `world` (tpc_mpc_comm_init_rank)."""
        buf = (C.c_char * capi.COMM_ID_BYTES).from_buffer_copy(bytes(comm_id).ljust(capi.COMM_ID_BYTES, b"\0"))
        self._check(self._lib.tpc_mpc_comm_init_rank(self._h, buf, capi.COMM_ID_BYTES, int(rank), int(world)))
        self.rank, self.world = int(rank), int(world)

    def solve_batch_compact_sharded(self, n_total: int, v_shard, dy_shard, dphi_shard, out=None,
                                    want_iters: bool = False, want_flags: bool = False, split: str = "block", **over):
        """This rank's shard of a batch of n_total, then the all-gather of the control outputs over RCCL
        (tpc_mpc_solve_batch_compact_sharded_split).  split "block": the shard is the contiguous block of shard_range;
        "interleaved": instances rank, rank + world, ... compacted (shard_map).  Device tensors; returns full-size
        (front, rear) in instance order under either split."""
        import torch
        p = self._params(**over)
        tdt = torch.float64 if p.dtype == capi.F64 else torch.float32
        first, count, stride = C.c_int64(), C.c_int64(), C.c_int64()
        self._lib.tpc_mpc_shard_map(int(n_total), self.rank, self.world, capi.SPLITS[split], C.byref(first), C.byref(count),
                                    C.byref(stride))
        for t in (v_shard, dy_shard, dphi_shard):
            if not (t.is_cuda and t.dtype == tdt and t.is_contiguous() and t.numel() == count.value):
                raise ValueError(f"shard arrays must be contiguous CUDA tensors of the solver dtype holding this "
                                 f"rank's {count.value} instances (rank {self.rank} of {self.world}, n_total {n_total})")
        if out is None:
            front = torch.empty(n_total, dtype=tdt, device=v_shard.device)
            rear = torch.empty(n_total, dtype=tdt, device=v_shard.device)
        else:
            front, rear = out
        iters = torch.empty(v_shard.numel(), dtype=torch.int32, device=v_shard.device) if want_iters else None
        flags = C.c_uint32(0)
        stream = torch.cuda.current_stream(v_shard.device).cuda_stream
        self._check(self._lib.tpc_mpc_solve_batch_compact_sharded_split(
            self._h, C.byref(p), int(n_total), capi.SPLITS[split], v_shard.data_ptr(), dy_shard.data_ptr(), dphi_shard.data_ptr(),
            front.data_ptr(), rear.data_ptr(), iters.data_ptr() if want_iters else None,
            C.byref(flags) if want_flags else None, C.c_void_p(stream)))
        self.last_flags = flags.value
        return (front, rear, iters) if want_iters else (front, rear)

    def shard_range(self, n_total: int):
        """(first, count) of this rank's contiguous block of a batch of n_total (tpc_mpc_shard_range)."""
        first, count = C.c_int64(), C.c_int64()
        self._lib.tpc_mpc_shard_range(int(n_total), self.rank, self.world, C.byref(first), C.byref(count))
        return first.value, count.value

    def shard_map(self, n_total: int, split: str = "block"):
        """(first, count, stride): element j of this rank's shard is instance first + j * stride (tpc_mpc_shard_map)."""
        first, count, stride = C.c_int64(), C.c_int64(), C.c_int64()
        self._lib.tpc_mpc_shard_map(int(n_total), self.rank, self.world, capi.SPLITS[split], C.byref(first), C.byref(count),
                                    C.byref(stride))
        return first.value, count.value, stride.value

    def gather_shards(self, n_total: int, *rows, split: str = "block"):
        """The exchange by itself, for any other entry a host shards (mixed horizons, follow, rollout): `rows` are FULL-size
        1-D CUDA tensors (or the rows of 2-D ones) of n_total elements of 4 or 8 bytes each, of which this rank has written
        its block [first, first + count) -- afterwards every rank holds all of every row (tpc_mpc_gather_shards_split).
        split "interleaved": the rank's shard sits compacted in the row's first `count` elements instead."""
        import torch
        flat = []
        for t in rows:
            if not (t.is_cuda and t.is_contiguous() and t.shape[-1] == n_total and t.element_size() in (4, 8)):
                raise ValueError("rows must be contiguous CUDA tensors whose last dimension is n_total, 4- or 8-byte elements")
            flat.extend(t.reshape(-1, n_total).unbind(0))
        if not flat:
            return
        es = flat[0].element_size()
        if any(r.element_size() != es for r in flat):
            raise ValueError("one call exchanges rows of one element size")
        table = (C.c_void_p * len(flat))(*[r.data_ptr() for r in flat])
        stream = torch.cuda.current_stream(flat[0].device).cuda_stream
        self._check(self._lib.tpc_mpc_gather_shards_split(self._h, int(n_total), capi.SPLITS[split], table, len(flat), es,
                                                          C.c_void_p(stream)))

    def solve_batch_general_sharded(self, A, B, Cc, Q, R, lower, upper, x0, targets, inputs: Optional[int] = None,
                                    want_iters: bool = False, **over):
        """The general form sharded (tpc_mpc_solve_batch_general_sharded): FULL-size component-major CUDA tensors
        [rows, n_total] on every rank (only this rank's block of columns has to be filled); solves that block in
        place and all-gathers the rows of u0 over RCCL.  Returns the full u0[I, n_total] (and this rank's iters)."""
        import torch
        p = self._params(**over)
        H = p.horizon
        tdt = torch.float64 if p.dtype == capi.F64 else torch.float32
        n = A.shape[-1]
        I = inputs or R.shape[0]

        def ptr(t, rows):
            if not (t.is_cuda and t.dtype == tdt and t.is_contiguous() and tuple(t.shape) == (rows, n)):
                raise ValueError(f"expected contiguous CUDA tensor [{rows},{n}] of the solver dtype")
            return t.data_ptr()
        u0 = torch.empty((I, n), dtype=tdt, device=A.device)
        iters = torch.zeros(n, dtype=torch.int32, device=A.device) if want_iters else None
        io = capi.GeneralIO(inputs=I, n=n, ld=n, A=ptr(A, 4), B=ptr(B, 2 * I), C=ptr(Cc, 2), Q=ptr(Q, 2),
                            R=ptr(R, I), lower=ptr(lower, I), upper=ptr(upper, I), x0=ptr(x0, 2),
                            targets=ptr(targets, 2 * H), controls_inout=None, v_inout=None, u0=u0.data_ptr(),
                            iters=iters.data_ptr() if want_iters else None)
        flags = C.c_uint32(0)
        stream = torch.cuda.current_stream(A.device).cuda_stream
        self._check(self._lib.tpc_mpc_solve_batch_general_sharded(self._h, C.byref(p), C.byref(io), C.byref(flags),
                                                                  C.c_void_p(stream)))
        self.last_flags = flags.value
        return (u0, iters) if want_iters else u0

    def solve_batch_general(self, A, B, Cc, Q, R, lower, upper, x0, targets, controls=None,
                            v_state=None, inputs: Optional[int] = None, want_iters: bool = False,
                            **over):
        """n independent dlib::mpc<2,I,H> controllers: ctor + set_target(t) + operator()(x0).

        Arrays are component-major (SoA): A[4,n] B[2I,n] C[2,n] Q[2,n] R[I,n] lower[I,n] upper[I,n]
        x0[2,n] targets[2H,n]; controls / v_state [H*I, n] are updated in place when given.
        Returns (u0[I,n][, iters])."""
        p = self._params(**over)
        H = p.horizon
        torch_mode = _is_torch(A)
        if torch_mode:
            import torch
            tdt = torch.float64 if p.dtype == capi.F64 else torch.float32
            n = A.shape[-1]
            I = inputs or R.shape[0]

            def ptr(t, rows):
                if t is None:
                    return None
                if not (t.is_cuda and t.dtype == tdt and t.is_contiguous() and tuple(t.shape) == (rows, n)):
                    raise ValueError(f"expected contiguous CUDA tensor [{rows},{n}] of the solver dtype")
                return t.data_ptr()
            u0 = torch.empty((I, n), dtype=tdt, device=A.device)
            iters = torch.empty(n, dtype=torch.int32, device=A.device) if want_iters else None
            stream = C.c_void_p(torch.cuda.current_stream(A.device).cuda_stream)
            mem = capi.DEVICE
            ip = iters.data_ptr() if want_iters else None
            up = u0.data_ptr()
        else:
            dt = _NP[p.dtype]
            A = np.ascontiguousarray(A, dtype=dt)
            n = A.shape[-1]
            I = inputs or np.asarray(R).shape[0]
            keep = []

            def ptr(a, rows):
                if a is None:
                    return None
                if not (isinstance(a, np.ndarray) and a.dtype == dt and a.flags.c_contiguous and a.shape == (rows, n)):
                    raise ValueError(f"expected C-contiguous ndarray [{rows},{n}] of the solver dtype")
                keep.append(a)
                return a.ctypes.data
            B, Cc, Q, R, lower, upper, x0, targets = (np.ascontiguousarray(a, dtype=dt) for a in
                                                      (B, Cc, Q, R, lower, upper, x0, targets))
            u0 = np.empty((I, n), dtype=dt)
            iters = np.empty(n, dtype=np.int32) if want_iters else None
            stream = None
            mem = capi.HOST
            ip = iters.ctypes.data if want_iters else None
            up = u0.ctypes.data
        io = capi.GeneralIO(inputs=I, n=n, ld=n, A=ptr(A, 4), B=ptr(B, 2 * I), C=ptr(Cc, 2), Q=ptr(Q, 2),
                            R=ptr(R, I), lower=ptr(lower, I), upper=ptr(upper, I), x0=ptr(x0, 2),
                            targets=ptr(targets, 2 * H), controls_inout=ptr(controls, H * I),
                            v_inout=ptr(v_state, H * I), u0=up, iters=ip)
        flags = C.c_uint32(0)
        self._check(self._lib.tpc_mpc_solve_batch_general(self._h, C.byref(p), C.byref(io),
                                                          C.byref(flags), mem, stream))
        self.last_flags = flags.value
        return (u0, iters) if want_iters else u0

    def rollout(self, steps: int, A, B, Cc, Q, R, lower, upper, x0, targets, new_last_targets=None,
                controls=None, v_state=None, inputs: Optional[int] = None, want_states: bool = True,
                want_iters: bool = False, **over):
        """`steps` successive operator() calls per controller with warm start and target shift
        (mpc.h:229-239) and the plant update of dlib/test/mpc.cpp:314 between them.  SoA arrays as
        in solve_batch_general; new_last_targets [2*steps, n].  Returns (controls[steps*I, n],
        states[steps*2, n] | None, iters[steps, n] | None)."""
        p = self._params(**over)
        H = p.horizon
        if _is_torch(A):
            import torch
            tdt = torch.float64 if p.dtype == capi.F64 else torch.float32
            n = A.shape[-1]
            I = inputs or R.shape[0]

            def ptr(t, rows):
                if t is None:
                    return None
                if not (t.is_cuda and t.dtype == tdt and t.is_contiguous() and tuple(t.shape) == (rows, n)):
                    raise ValueError(f"expected contiguous CUDA tensor [{rows},{n}] of the solver dtype")
                return t.data_ptr()
            c_out = torch.empty((steps * I, n), dtype=tdt, device=A.device)
            s_out = torch.empty((steps * 2, n), dtype=tdt, device=A.device) if want_states else None
            i_out = torch.empty((steps, n), dtype=torch.int32, device=A.device) if want_iters else None
            stream = C.c_void_p(torch.cuda.current_stream(A.device).cuda_stream)
            mem = capi.DEVICE
            optr = lambda t: None if t is None else t.data_ptr()
        else:
            dt = _NP[p.dtype]
            A = np.ascontiguousarray(A, dtype=dt)
            n = A.shape[-1]
            I = inputs or np.asarray(R).shape[0]
            keep = []

            def ptr(a, rows):
                if a is None:
                    return None
                if not (isinstance(a, np.ndarray) and a.dtype == dt and a.flags.c_contiguous and a.shape == (rows, n)):
                    raise ValueError(f"expected C-contiguous ndarray [{rows},{n}] of the solver dtype")
                keep.append(a)
                return a.ctypes.data
            B, Cc, Q, R, lower, upper, x0, targets = (np.ascontiguousarray(a, dtype=dt) for a in
                                                      (B, Cc, Q, R, lower, upper, x0, targets))
            if new_last_targets is not None:
                new_last_targets = np.ascontiguousarray(new_last_targets, dtype=dt)
            c_out = np.empty((steps * I, n), dtype=dt)
            s_out = np.empty((steps * 2, n), dtype=dt) if want_states else None
            i_out = np.empty((steps, n), dtype=np.int32) if want_iters else None
            stream = None
            mem = capi.HOST
            optr = lambda a: None if a is None else a.ctypes.data
        io = capi.GeneralIO(inputs=I, n=n, ld=n, A=ptr(A, 4), B=ptr(B, 2 * I), C=ptr(Cc, 2), Q=ptr(Q, 2),
                            R=ptr(R, I), lower=ptr(lower, I), upper=ptr(upper, I), x0=ptr(x0, 2),
                            targets=ptr(targets, 2 * H), controls_inout=ptr(controls, H * I),
                            v_inout=ptr(v_state, H * I), u0=None, iters=None)
        flags = C.c_uint32(0)
        self._check(self._lib.tpc_mpc_rollout(self._h, C.byref(p), C.byref(io), int(steps),
                                              ptr(new_last_targets, 2 * steps), optr(c_out), optr(s_out),
                                              optr(i_out), C.byref(flags), mem, stream))
        self.last_flags = flags.value
        return c_out, s_out, i_out

    def follow_batch(self, pos_x, pos_y, dir_x, dir_y, velocity, count, car_velocity, look_ahead,
                     lookup=None, want_iters: bool = False, **over):
        """Batched tobiMPC branch of cycle() on raw trajectories (device tensors).

        pos_x .. velocity: float32 CUDA tensors [max_points, n] (point-major, SoA); count int32 [n];
        car_velocity, look_ahead float32 [n]; lookup: optional (x, y) float32 CUDA tensors of the
        velocity lookup table.  Returns (steering_front f64, steering_rear f64, target_speed f32,
        target_distance f32[, iters])."""
        import torch
        p = self._params(**over)
        P, n = pos_x.shape
        dev = pos_x.device
        for tns in (pos_x, pos_y, dir_x, dir_y, velocity):
            if not (tns.is_cuda and tns.dtype == torch.float32 and tns.is_contiguous() and tuple(tns.shape) == (P, n)):
                raise ValueError("trajectory arrays must be contiguous float32 CUDA tensors [max_points, n]")
        for tns, dt in ((count, torch.int32), (car_velocity, torch.float32), (look_ahead, torch.float32)):
            if not (tns.is_cuda and tns.dtype == dt and tns.is_contiguous() and tns.numel() == n):
                raise ValueError("per-instance arrays must be contiguous CUDA tensors of length n")
        tr = capi.Trajectories(n=n, ld=n, max_points=P, pos_x=pos_x.data_ptr(), pos_y=pos_y.data_ptr(),
                               dir_x=dir_x.data_ptr(), dir_y=dir_y.data_ptr(), velocity=velocity.data_ptr(),
                               count=count.data_ptr(), car_velocity=car_velocity.data_ptr(),
                               look_ahead=look_ahead.data_ptr())
        front = torch.empty(n, dtype=torch.float64, device=dev)
        rear = torch.empty(n, dtype=torch.float64, device=dev)
        tspeed = torch.empty(n, dtype=torch.float32, device=dev)
        tdist = torch.empty(n, dtype=torch.float32, device=dev)
        iters = torch.empty(n, dtype=torch.int32, device=dev) if want_iters else None
        lx, ly, ln = (None, None, 0) if lookup is None else (lookup[0].data_ptr(), lookup[1].data_ptr(), lookup[0].numel())
        flags = C.c_uint32(0)
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        self._check(self._lib.tpc_mpc_follow_batch(self._h, C.byref(p), C.byref(tr), lx, ly, ln,
                                                   front.data_ptr(), rear.data_ptr(), tspeed.data_ptr(),
                                                   tdist.data_ptr(), iters.data_ptr() if want_iters else None,
                                                   C.byref(flags), stream))
        self.last_flags = flags.value
        return (front, rear, tspeed, tdist, iters) if want_iters else (front, rear, tspeed, tdist)

    def follow_batch_horizon(self, pos_x, pos_y, dir_x, dir_y, velocity, count, car_velocity, look_ahead,
                             step_spacing=None, lookup=None, want_iters: bool = False,
                             want_targets: bool = False, **over):
        """follow_batch with one trajectory point per horizon step (tpc_mpc_follow_batch_horizon):
        step t's target is the polyline point at arc length look_ahead + t * spacing.  step_spacing:
        optional float32 CUDA tensor [n] (None: |v| * step_size).  Returns (front, rear, target_speed,
        target_distance[, targets f64 [2H, n]][, iters])."""
        import torch
        p = self._params(**over)
        P, n = pos_x.shape
        dev = pos_x.device
        for tns in (pos_x, pos_y, dir_x, dir_y, velocity):
            if not (tns.is_cuda and tns.dtype == torch.float32 and tns.is_contiguous() and tuple(tns.shape) == (P, n)):
                raise ValueError("trajectory arrays must be contiguous float32 CUDA tensors [max_points, n]")
        per = [(count, torch.int32), (car_velocity, torch.float32), (look_ahead, torch.float32)]
        if step_spacing is not None:
            per.append((step_spacing, torch.float32))
        for tns, dt in per:
            if not (tns.is_cuda and tns.dtype == dt and tns.is_contiguous() and tns.numel() == n):
                raise ValueError("per-instance arrays must be contiguous CUDA tensors of length n")
        tr = capi.Trajectories(n=n, ld=n, max_points=P, pos_x=pos_x.data_ptr(), pos_y=pos_y.data_ptr(),
                               dir_x=dir_x.data_ptr(), dir_y=dir_y.data_ptr(), velocity=velocity.data_ptr(),
                               count=count.data_ptr(), car_velocity=car_velocity.data_ptr(),
                               look_ahead=look_ahead.data_ptr())
        front = torch.empty(n, dtype=torch.float64, device=dev)
        rear = torch.empty(n, dtype=torch.float64, device=dev)
        tspeed = torch.empty(n, dtype=torch.float32, device=dev)
        tdist = torch.empty(n, dtype=torch.float32, device=dev)
        targets = torch.empty((2 * p.horizon, n), dtype=torch.float64, device=dev) if want_targets else None
        iters = torch.empty(n, dtype=torch.int32, device=dev) if want_iters else None
        lx, ly, ln = (None, None, 0) if lookup is None else (lookup[0].data_ptr(), lookup[1].data_ptr(), lookup[0].numel())
        flags = C.c_uint32(0)
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        self._check(self._lib.tpc_mpc_follow_batch_horizon(
            self._h, C.byref(p), C.byref(tr), step_spacing.data_ptr() if step_spacing is not None else None,
            lx, ly, ln, front.data_ptr(), rear.data_ptr(), tspeed.data_ptr(), tdist.data_ptr(),
            targets.data_ptr() if want_targets else None, iters.data_ptr() if want_iters else None,
            C.byref(flags), stream))
        self.last_flags = flags.value
        out = [front, rear, tspeed, tdist]
        if want_targets:
            out.append(targets)
        if want_iters:
            out.append(iters)
        return tuple(out)
