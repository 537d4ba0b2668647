import sys; sys.path.insert(0, ".")
import torch
from trajectory_controller_amd import MpcSolver
from trajectory_controller_amd.synth import compact_inputs
for H in (4, 5, 10):
    for n in (262144, 1048576):
        v, dy, dphi = (torch.from_numpy(a).cuda() for a in compact_inputs(H, n))
        with MpcSolver(horizon=H, algo="lane") as s:
            s.set_profiling(True)
            for _ in range(3):
                f, r, it = s.solve_batch_compact(v, dy, dphi, want_iters=True, want_flags=False)
                k1, k2, _ = s.last_kernel_times()
            wi, rb = s.last_lane_stats()
            pg = (it.double() - 50).clamp(min=0)
            print(f"H={H} n={n}: cd {k1:.3f} ms pg {k2:.3f} ms; wave_iters {wi} refill passes {rb}; instances needing PG {int((pg>0).sum())}; "
                  f"sum PG iters {int(pg.sum())}; per wave (2048): iters {wi/2048:.0f} passes {rb/2048:.1f}; us per wave-iter if all time were iterations {k2*1e3/(wi/2048):.3f}")
