"""trajectory_controller_amd -- MI355X-native batched MPC-QP solver behind the
TrajectoryPointController module surface of lms-org/trajectory_controller.

Only the hot path lives here: csrc/ holds the gfx950 HIP kernels and the C ABI
(include/tpc_mpc.h -> lib/libtpc_mpc.so); capi.py / solver.py are the host-side binding;
host/ is the C++ module shim that keeps the LMS surface.  Nothing in this package imports
oracle/ -- that directory is the checker used by tests/ and bench.py only.
"""
from .capi import (ALGO_AUTO, ALGO_GROUP, ALGO_LANE, ALGO_LANE_FMA, ALGO_WAVE, F32, F64, FLAG_BAD_MODEL, FLAG_MAX_ITER, FLAG_NONFINITE,
                   TpcMpcError, default_params, load_library)
from .solver import MpcSolver

__all__ = ["MpcSolver", "TpcMpcError", "default_params", "load_library", "ALGO_AUTO", "ALGO_WAVE",
           "ALGO_LANE", "ALGO_LANE_FMA", "ALGO_GROUP", "F64", "F32", "FLAG_NONFINITE", "FLAG_MAX_ITER", "FLAG_BAD_MODEL"]
