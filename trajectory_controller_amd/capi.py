"""ctypes binding of the C-ABI product library (include/tpc_mpc.h -> lib/libtpc_mpc.so).

This is the same binding a non-Python host would write (INTEGRATION.md shows the C++ one).  The
library is built in-tree by `__graft_entry__.build()` / `make -C trajectory_controller_amd/csrc`.
There is no fallback: if the library is missing, loading raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# TPC_MPC_LIB lets an A/B run load another build of the same ABI (tuning only)
LIB_PATH = os.environ.get("TPC_MPC_LIB") or os.path.join(_HERE, "lib", "libtpc_mpc.so")

OK = 0
F64, F32 = 0, 1
HOST, DEVICE = 0, 1
ALGO_AUTO, ALGO_WAVE, ALGO_LANE, ALGO_LANE_FMA, ALGO_GROUP = 0, 1, 2, 3, 4
OPT_WAVE_GROUP, OPT_MAILBOX_HOST, OPT_GROUP_LANES, OPT_HOST_SOLVE_ONE = 1, 2, 3, 4
DEVICE_NONE = -1   # tpc_mpc_create: a host-only handle
FLAG_NONFINITE, FLAG_MAX_ITER, FLAG_BAD_MODEL = 0x1, 0x2, 0x4
PARAM_FAST_CAPPED = 0x1   # tpc_mpc_params.options

STATUS_NAMES = {0: "OK", 1: "BAD_ARG", 2: "BAD_WEIGHTS", 3: "BAD_BOUNDS", 4: "BAD_HORIZON",
                5: "BAD_EPS", 6: "NO_DEVICE", 7: "HIP", 8: "ALLOC", 9: "COMM"}
ABI_VERSION = 5
COMM_ID_BYTES = 128

# every symbol include/tpc_mpc.h declares
EXPORTS = ("tpc_mpc_default_params", "tpc_mpc_create", "tpc_mpc_destroy", "tpc_mpc_last_error",
           "tpc_mpc_supported_horizons", "tpc_mpc_abi_version", "tpc_mpc_solve_one",
           "tpc_mpc_solve_batch_compact", "tpc_mpc_solve_batch_general", "tpc_mpc_rollout",
           "tpc_mpc_set_profiling", "tpc_mpc_last_kernel_times", "tpc_mpc_last_lane_stats",
           "tpc_mpc_follow_batch", "tpc_mpc_set_option", "tpc_mpc_reserve",
           "tpc_mpc_build_info", "tpc_mpc_set_resident", "tpc_mpc_solve_batch_compact_mixed",
           "tpc_mpc_follow_batch_horizon", "tpc_mpc_comm_unique_id", "tpc_mpc_comm_init_rank",
           "tpc_mpc_comm_destroy", "tpc_mpc_group_begin", "tpc_mpc_group_end", "tpc_mpc_shard_range",
           "tpc_mpc_solve_batch_compact_sharded", "tpc_mpc_comm_test_mode",
           "tpc_mpc_solve_batch_general_sharded", "tpc_mpc_last_flags", "tpc_mpc_gather_shards",
           "tpc_mpc_shard_map", "tpc_mpc_solve_batch_compact_sharded_split", "tpc_mpc_gather_shards_split")
SPLIT_BLOCK, SPLIT_INTERLEAVED = 0, 1
SPLITS = {"block": SPLIT_BLOCK, "interleaved": SPLIT_INTERLEAVED}


class Params(C.Structure):
    """struct tpc_mpc_params"""
    _fields_ = [("horizon", C.c_int32), ("dtype", C.c_int32), ("algo", C.c_int32),
                ("options", C.c_int32), ("eps", C.c_double), ("max_iter", C.c_uint64),
                ("smo_iters", C.c_uint64), ("step_size", C.c_double), ("wheelbase", C.c_double),
                ("weight_y", C.c_double), ("weight_phi", C.c_double),
                ("weight_steering_front", C.c_double), ("weight_steering_rear", C.c_double),
                ("lower", C.c_double * 2), ("upper", C.c_double * 2)]


class GeneralIO(C.Structure):
    """struct tpc_mpc_general_io"""
    _fields_ = [("inputs", C.c_int32), ("reserved", C.c_int32), ("n", C.c_int64), ("ld", C.c_int64),
                ("A", C.c_void_p), ("B", C.c_void_p), ("C", C.c_void_p), ("Q", C.c_void_p),
                ("R", C.c_void_p), ("lower", C.c_void_p), ("upper", C.c_void_p),
                ("x0", C.c_void_p), ("targets", C.c_void_p), ("controls_inout", C.c_void_p),
                ("v_inout", C.c_void_p), ("u0", C.c_void_p), ("iters", C.c_void_p)]


class Trajectories(C.Structure):
    """struct tpc_mpc_trajectories"""
    _fields_ = [("n", C.c_int64), ("ld", C.c_int64), ("max_points", C.c_int32), ("reserved", C.c_int32),
                ("pos_x", C.c_void_p), ("pos_y", C.c_void_p), ("dir_x", C.c_void_p), ("dir_y", C.c_void_p),
                ("velocity", C.c_void_p), ("count", C.c_void_p), ("car_velocity", C.c_void_p),
                ("look_ahead", C.c_void_p)]


class TpcMpcError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(f"tpc_mpc status {status} ({STATUS_NAMES.get(status, '?')}): {message}")
        self.status = status


_lib = None


def load_library(path: str | None = None) -> C.CDLL:
    """dlopen libtpc_mpc.so and declare its prototypes.  Raises if the library was not built."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    path = path or LIB_PATH
    if not os.path.exists(path):
        raise FileNotFoundError(
            f"{path} not found: build the HIP library first "
            f"(python -c 'import __graft_entry__ as g; g.build()' or make -C trajectory_controller_amd/csrc)")
    # One HIP runtime per process.  The PyTorch wheel bundles its own libamdhip64 / libhsa-runtime64; libtpc_mpc.so names
    # them by SONAME only, so it binds to whichever copy is already loaded.  Loaded FIRST it pulls in /opt/rocm's, torch
    # then brings its own, and the second runtime to initialise finds no device ("no ROCm-capable device is detected";
    # measured: scripts/probes/import_order.py).  A Python process that has torch gets torch's runtime loaded first here,
    # whatever the import order of the caller; a process without torch (a C host) has only /opt/rocm's anyway.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(path)
    vp, i32p, u32p = C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_uint32)
    lib.tpc_mpc_abi_version.restype = C.c_int
    lib.tpc_mpc_supported_horizons.argtypes = [C.POINTER(C.c_int), C.c_int]
    lib.tpc_mpc_default_params.argtypes = [C.POINTER(Params), C.c_int]
    lib.tpc_mpc_create.argtypes = [C.c_int, C.POINTER(vp)]
    lib.tpc_mpc_destroy.argtypes = [vp]
    lib.tpc_mpc_last_error.argtypes = [vp]
    lib.tpc_mpc_last_error.restype = C.c_char_p
    lib.tpc_mpc_solve_one.argtypes = [vp, C.POINTER(Params), C.c_double, C.c_double, C.c_double,
                                      C.POINTER(C.c_double), C.POINTER(C.c_double)]
    lib.tpc_mpc_last_flags.argtypes = [vp, u32p, i32p]
    lib.tpc_mpc_solve_batch_compact.argtypes = [vp, C.POINTER(Params), C.c_int64, vp, vp, vp, vp, vp,
                                                vp, u32p, C.c_int, vp]
    lib.tpc_mpc_solve_batch_general.argtypes = [vp, C.POINTER(Params), C.POINTER(GeneralIO), u32p,
                                                C.c_int, vp]
    lib.tpc_mpc_solve_batch_general_sharded.argtypes = [vp, C.POINTER(Params), C.POINTER(GeneralIO), u32p, vp]
    lib.tpc_mpc_rollout.argtypes = [vp, C.POINTER(Params), C.POINTER(GeneralIO), C.c_int32, vp, vp, vp,
                                    vp, u32p, C.c_int, vp]
    lib.tpc_mpc_set_profiling.argtypes = [vp, C.c_int]
    lib.tpc_mpc_last_kernel_times.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                              C.POINTER(C.c_int)]
    lib.tpc_mpc_follow_batch.argtypes = [vp, C.POINTER(Params), C.POINTER(Trajectories), vp, vp, C.c_int32,
                                         vp, vp, vp, vp, vp, u32p, vp]
    lib.tpc_mpc_last_lane_stats.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    lib.tpc_mpc_x_set_work_hint.argtypes = [vp, vp, C.c_int64, C.c_int]   # experimental (csrc/tpc_mpc_experimental.h)
    lib.tpc_mpc_set_option.argtypes = [vp, C.c_int, C.c_int64]
    lib.tpc_mpc_x_set_group_share.argtypes = [vp, C.c_int, C.c_int]       # experimental
    lib.tpc_mpc_x_set_lanex_below.argtypes = [vp, C.c_int64]               # experimental
    lib.tpc_mpc_gather_shards.argtypes = [vp, C.c_int64, C.POINTER(C.c_void_p), C.c_int, C.c_int, vp]
    lib.tpc_mpc_reserve.argtypes = [vp, C.POINTER(Params), C.c_int64, C.c_int]
    lib.tpc_mpc_build_info.restype = C.c_char_p
    lib.tpc_mpc_set_resident.argtypes = [vp, C.c_int64]
    lib.tpc_mpc_solve_batch_compact_mixed.argtypes = [vp, C.POINTER(Params), C.c_int64, vp, vp, vp, vp, vp, vp,
                                                      vp, u32p, C.c_int, vp]
    lib.tpc_mpc_follow_batch_horizon.argtypes = [vp, C.POINTER(Params), C.POINTER(Trajectories), vp, vp, vp,
                                                 C.c_int32, vp, vp, vp, vp, vp, vp, u32p, vp]
    lib.tpc_mpc_comm_unique_id.argtypes = [vp, C.c_size_t]
    lib.tpc_mpc_comm_init_rank.argtypes = [vp, vp, C.c_size_t, C.c_int, C.c_int]
    lib.tpc_mpc_comm_test_mode.argtypes = [vp, C.c_int, C.c_int]
    lib.tpc_mpc_comm_destroy.argtypes = [vp]
    lib.tpc_mpc_shard_range.argtypes = [C.c_int64, C.c_int, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    lib.tpc_mpc_solve_batch_compact_sharded.argtypes = [vp, C.POINTER(Params), C.c_int64, vp, vp, vp, vp, vp, vp,
                                                        u32p, vp]
    i64p = C.POINTER(C.c_int64)
    lib.tpc_mpc_shard_map.argtypes = [C.c_int64, C.c_int, C.c_int, C.c_int, i64p, i64p, i64p]
    lib.tpc_mpc_solve_batch_compact_sharded_split.argtypes = [vp, C.POINTER(Params), C.c_int64, C.c_int, vp, vp, vp, vp, vp, vp,
                                                              u32p, vp]
    lib.tpc_mpc_gather_shards_split.argtypes = [vp, C.c_int64, C.c_int, C.POINTER(C.c_void_p), C.c_int, C.c_int, vp]
    lib.tpc_mpc_x_exchange_plan.argtypes = [C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, i64p, C.c_int, i64p]   # experimental
    for name in EXPORTS:
        getattr(lib, name)   # raises AttributeError if the library lacks a declared entry point
    if path == LIB_PATH:
        _lib = lib
    return lib


def default_params(horizon: int, dtype: int = F64, algo: int = ALGO_AUTO, **overrides) -> Params:
    lib = load_library()
    p = Params()
    rc = lib.tpc_mpc_default_params(C.byref(p), horizon)
    if rc != OK:
        raise TpcMpcError(rc, f"unsupported horizon {horizon}")
    p.dtype, p.algo = dtype, algo
    for k, val in overrides.items():
        if k in ("lower", "upper"):
            getattr(p, k)[0], getattr(p, k)[1] = float(val[0]), float(val[1])
        else:
            setattr(p, k, val)
    return p
