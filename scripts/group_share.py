#!/usr/bin/env python3
"""Diagnostic: one GROUP batch on a SHARE of the chip (what a bin of a mixed-horizon batch gets):
    python scripts/group_share.py H G n waves[,waves...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from trajectory_controller_amd import MpcSolver, capi
from trajectory_controller_amd.synth import compact_inputs
H, G, n = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
tv, ty, tp = (torch.from_numpy(a).cuda() for a in compact_inputs(H, n))
for w in [int(x) for x in sys.argv[4].split(",")]:
    with MpcSolver(horizon=H, algo="group") as s:
        s.set_option(capi.OPT_GROUP_LANES, G)
        s._check(s._lib.tpc_mpc_x_set_group_share(s._h, w, 0))
        s.set_profiling(True)
        best = 1e9
        for _ in range(3):
            s.solve_batch_compact(tv, ty, tp, want_flags=False)
            k1, k2, _ = s.last_kernel_times()
            best = min(best, k2)
        wi, rb = s.last_lane_stats()
        print(f"H={H} G={G} n={n} waves={w}: pg {best:.3f} ms  wave-iters {wi}  per wave {wi / max(1, (w or 1024)):.0f}  -> {best * 1e3 / (wi / max(1, (w or 1024))):.3f} us per wave-iteration", flush=True)
