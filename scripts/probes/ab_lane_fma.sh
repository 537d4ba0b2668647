#!/bin/bash
# interleaved A/B of builds on one box, the compiler-built LANE_FMA kernels: scripts/probes/ab_lane_fma.sh ROUNDS dir...
R=$1; shift
for i in $(seq $R); do
  for L in "$@"; do
    echo "== $L"
    TPC_MPC_LIB=$PWD/$L/libtpc_mpc.so timeout -k 10 300 python - <<'PY' 2>&1 | grep -v amdgpu
import time, torch
from trajectory_controller_amd import MpcSolver
from trajectory_controller_amd.synth import compact_inputs
for H, n, dt in ((10, 262144, "f64"), (10, 1048576, "f64"), (20, 262144, "f32"), (20, 2097152, "f32"), (40, 262144, "f32"), (5, 1048576, "f64")):
    v, dy, dphi = (torch.from_numpy(a).cuda() for a in compact_inputs(H, n))
    if dt == "f32": v, dy, dphi = v.float(), dy.float(), dphi.float()
    with MpcSolver(horizon=H, algo="lane_fma", dtype=dt) as s:
        s.set_profiling(True)
        s.solve_batch_compact(v, dy, dphi); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): s.solve_batch_compact(v, dy, dphi)
        torch.cuda.synchronize(); t = (time.perf_counter()-t0)/5*1e3
        print(f"LANE_FMA {dt} {n} x N={H}: {t:.3f} ms  (PG kernel {s.last_kernel_times()[1]:.3f})")
PY
  done
done
