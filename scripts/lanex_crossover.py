#!/usr/bin/env python3
"""The bit-exact LANE family, compact form, fp64: one lane per instance (lane_pg_fused_kernel) against G lanes per instance
(lanex_pg_kernel, csrc/mpc_lanex.h) over batch size -- the crossover behind lanex_below() in csrc/mpc_lane_inst.hip -- and
that the two give the same bits:
    python scripts/lanex_crossover.py [H,H,...] [n,n,...] [general I]     (general: the general model with I inputs)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from trajectory_controller_amd import MpcSolver
from trajectory_controller_amd.synth import compact_inputs, general_inputs
import numpy as np
GEN = int(sys.argv[4]) if len(sys.argv) > 4 and sys.argv[3] == "general" else 0
hs = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [10, 20, 40]
ns = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1024, 4096, 8192, 16384, 24576, 32768, 49152, 65536, 131072]
for H in hs:
    cross = None
    for n in ns:
        if GEN:
            g = general_inputs(H, n, I=GEN)
            gdev = [torch.from_numpy(np.ascontiguousarray(g[k].reshape(n, -1).T)).cuda() for k in ("A", "B", "C", "Q", "R", "lo", "hi", "x0", "targets")]
        else:
            tv, ty, tp = (torch.from_numpy(a).cuda() for a in compact_inputs(H, n))
        res = {}
        for name, below in (("lane", 0), ("lanex", 1 << 40)):
            with MpcSolver(horizon=H, algo="lane") as s:
                s._check(s._lib.tpc_mpc_x_set_lanex_below(s._h, below))
                s.set_profiling(True)
                best = 1e9
                for _ in range(3):
                    if GEN:
                        f, it = s.solve_batch_general(*gdev, inputs=GEN, want_iters=True)
                        r = f
                    else:
                        f, r, it = s.solve_batch_compact(tv, ty, tp, want_iters=True, want_flags=False)
                    k1, k2, _ = s.last_kernel_times()
                    best = min(best, k1 + k2)
                res[name] = (best, f.clone(), r.clone(), it.clone())
        same = bool(torch.equal(res["lane"][1].view(torch.int64), res["lanex"][1].view(torch.int64)) and
                    torch.equal(res["lane"][2].view(torch.int64), res["lanex"][2].view(torch.int64)) and
                    torch.equal(res["lane"][3], res["lanex"][3]))
        tl, tx = res["lane"][0], res["lanex"][0]
        if cross is None and tx > tl:
            cross = n
        print(f"f64 {'general I=' + str(GEN) + ' ' if GEN else ''}H={H:2d} n={n:6d}: one lane per instance {tl:8.3f} ms   G lanes {tx:8.3f} ms   bits and iteration counts equal: {same}", flush=True)
    print(f"   => H={H}: G lanes per instance faster below n = {cross}", flush=True)
