import os, sys
sys.path.insert(0, '.')
import torch
from trajectory_controller_amd import MpcSolver
from trajectory_controller_amd.synth import compact_inputs
for H in (4, 5, 10):
    n = 1 << 20
    v, dy, dphi = (torch.from_numpy(a).cuda() for a in compact_inputs(H, n))
    with MpcSolver(horizon=H, algo="lane") as s:
        s.set_profiling(True)
        for _ in range(3):
            s.solve_batch_compact(v, dy, dphi, want_flags=False)
            k1, k2, _ = s.last_kernel_times()
    print(f"lane f64 H={H} n={n}: cd {k1:.3f} ms pg {k2:.3f} ms  {n/(k1+k2)/1e3:.1f} Msolve/s", flush=True)
