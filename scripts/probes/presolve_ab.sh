#!/bin/bash
# AUTO's presolve: config 5 as shipped and the single N = 40 bin, per build (scripts/probes/presolve_ab.sh dir...)
for L in "$@"; do
  echo "== $L"
  TPC_MPC_LIB=$PWD/$L/libtpc_mpc.so timeout -k 10 120 python scripts/mixed_horizons.py f64 f64fast 2>&1 | grep "ms per"
  TPC_MPC_LIB=$PWD/$L/libtpc_mpc.so timeout -k 10 120 python - <<'PY' 2>&1 | grep -v amdgpu
import time, torch, numpy as np
from trajectory_controller_amd import MpcSolver
from trajectory_controller_amd.synth import compact_inputs
for H, n in ((40, 16384), (40, 65536), (30, 16384)):
    v, dy, dphi = (torch.from_numpy(a).cuda() for a in compact_inputs(H, n))
    for algo in ("auto", "lane"):
        with MpcSolver(horizon=H, algo=algo) as s:
            s.solve_batch_compact(v, dy, dphi); torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(3): s.solve_batch_compact(v, dy, dphi)
            torch.cuda.synchronize(); print(f"compact {n} x N={H} {algo}: {(time.perf_counter()-t0)/3*1e3:.2f} ms")
PY
done
