#!/bin/bash
# the headline kernel's rate over the batch size on one box (the launch tail): scripts/probes/tail_sizes.sh
for n in 262144 524288 1048576 2097152; do
  timeout -k 10 300 python - $n <<'PY' 2>&1 | grep -v amdgpu
import sys, time, torch
from trajectory_controller_amd import MpcSolver
from trajectory_controller_amd.synth import compact_inputs
n = int(sys.argv[1]); H = 20
v, dy, dphi = (torch.from_numpy(a).cuda() for a in compact_inputs(H, n))
front, rear = torch.empty_like(v), torch.empty_like(v)
with MpcSolver(horizon=H, algo="auto") as s:   # (as bench.py's step: outputs preallocated, no flag read-back)
    s.set_profiling(True); s.reserve(n)
    for _ in range(3): s.solve_batch_compact(v, dy, dphi, out=(front, rear), want_flags=False)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): s.solve_batch_compact(v, dy, dphi, out=(front, rear), want_flags=False)
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 10
    k1, k2, algo = s.last_kernel_times(); wi, rf = s.last_lane_stats()
    print(f"n={n}: {n/t/1e6:.2f} M solves/s, {t*1e3:.3f} ms per step; CD {k1:.3f} ms, PG {k2:.3f} ms = {n/k2/1e3:.2f} M/s kernel-only; wave iterations per wavefront {wi/1024:.0f}")
PY
done
