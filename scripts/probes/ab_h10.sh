#!/bin/bash
# interleaved A/B on one box, LANE_FMA fp64 N = 10 (the hand-written kernel against the compiler's: ab/noasm10 =
# scripts/build_ub_variant.sh noasm10 10 "-DTPC_UB_NO_ASM"): scripts/probes/ab_h10.sh ROUNDS dir...
R=$1; shift
for i in $(seq $R); do
  for L in "$@"; do
    echo "== $L"
    TPC_MPC_LIB=$PWD/$L/libtpc_mpc.so timeout -k 10 200 python - <<'PY' 2>&1 | grep -v amdgpu
import time, torch
from trajectory_controller_amd import MpcSolver
from trajectory_controller_amd.synth import compact_inputs
H = 10
for n in (131072, 262144, 1048576):
    v, dy, dphi = (torch.from_numpy(a).cuda() for a in compact_inputs(H, n))
    front, rear = torch.empty_like(v), torch.empty_like(v)
    with MpcSolver(horizon=H, algo="lane_fma") as s:
        s.set_profiling(True); s.reserve(n)
        for _ in range(3): s.solve_batch_compact(v, dy, dphi, out=(front, rear), want_flags=False)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): s.solve_batch_compact(v, dy, dphi, out=(front, rear), want_flags=False)
        torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 10
        k1, k2, _ = s.last_kernel_times(); wi, rf = s.last_lane_stats()
        print(f"N=10 n={n}: {n/t/1e6:.1f} M solves/s, {t*1e3:.3f} ms per step (CD {k1:.3f}, PG {k2:.3f}); wave iterations {wi}, refill passes {rf}, PG ns per wave iteration x SIMD {k2*1e6*1024/max(wi,1):.0f}")
PY
  done
done
