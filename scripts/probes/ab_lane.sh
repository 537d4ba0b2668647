#!/bin/bash
# interleaved A/B of builds on one box, the bit-exact LANE family: scripts/probes/ab_lane.sh ROUNDS dir...  (262 144 x N=20 and N=10, fp64 and fp32)
R=$1; shift
for i in $(seq $R); do
  for L in "$@"; do
    echo "== $L"
    TPC_MPC_LIB=$PWD/$L/libtpc_mpc.so timeout -k 10 200 python - <<'PY' 2>&1 | grep -v amdgpu
import time, torch
from trajectory_controller_amd import MpcSolver
from trajectory_controller_amd.synth import compact_inputs
for H, n, dt in ((20, 262144, "f64"), (10, 262144, "f64"), (40, 65536, "f64"), (20, 262144, "f32")):
    v, dy, dphi = (torch.from_numpy(a).cuda() for a in compact_inputs(H, n))
    if dt == "f32": v, dy, dphi = v.float(), dy.float(), dphi.float()
    with MpcSolver(horizon=H, algo="lane", dtype=dt) as s:
        s.solve_batch_compact(v, dy, dphi); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): s.solve_batch_compact(v, dy, dphi)
        torch.cuda.synchronize(); print(f"LANE {dt} {n} x N={H}: {(time.perf_counter()-t0)/5*1e3:.3f} ms")
PY
  done
done
