#!/usr/bin/env python3
"""Probe (not product): is a SPLIT launch worth building for a latency-bound fp32 batch?

A 262 144-instance fp32 batch at N = 20 lasts as long as its longest instances (NOTEBOOK.md, round 4): LANE_FMA's steady
state is 87 M solves/s, the batch gets 57 M from it and 63 M from GROUP (two lanes per instance, two wavefronts per SIMD).
The split: the fastest-moving share of the instances (iteration counts follow the speed: correlation 0.94) through GROUP
on one stream, the rest through LANE_FMA on another, two handles, both persistent grids resident at once -- GROUP's pinned
to one wavefront per SIMD (tpc_mpc_x_set_group_share) and launched first, so that LANE_FMA's two-wavefront workgroups
take the other half of every CU's slots.  The partition by speed is done here with torch (a product version would bin on
the device the way tpc_mpc_solve_batch_compact_mixed bins by horizon); its cost is reported separately.

    python scripts/probes/split_launch_f32.py [n] [H]
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from trajectory_controller_amd import MpcSolver, capi
from trajectory_controller_amd.synth import compact_inputs

n = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
H = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda", 0)
cus = torch.cuda.get_device_properties(0).multi_processor_count
tv, ty, tp = (torch.from_numpy(a).to(dev, dtype=torch.float32) for a in compact_inputs(H, n))


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def whole(algo, G=0):
    s = MpcSolver(horizon=H, dtype="f32", algo=algo)
    if G:
        s.set_option(capi.OPT_GROUP_LANES, G)
    s.reserve(n)
    f, r = torch.empty_like(tv), torch.empty_like(tv)
    ms = timed(lambda: s.solve_batch_compact(tv, ty, tp, out=(f, r), want_flags=False))
    s.close()
    return ms, f, r


base = {}
for name, algo, G in (("lane_fma", "lane_fma", 0), ("group G=2", "group", 2), ("group G=4", "group", 4), ("auto", "auto", 0)):
    base[name] = whole(algo, G)
    print(f"whole batch, {name:10s}: {base[name][0]:7.3f} ms = {n / base[name][0] / 1e3:6.1f} M solves/s", flush=True)

order = torch.argsort(tv, descending=True)
t_part = timed(lambda: (torch.argsort(tv, descending=True), tv[order], ty[order], tp[order]))
print(f"(partition by speed with torch: argsort + three gathers {t_part:.3f} ms -- a device-side binning pass would be ~0.05 ms)")
sv, sy, sp = tv[order].contiguous(), ty[order].contiguous(), tp[order].contiguous()
for phi in (0.1, 0.15, 0.2, 0.25, 0.3, 0.4):
    for G, per_simd in ((2, 1), (4, 1), (2, 2), (4, 2)):
        k = int(phi * n)
        a = MpcSolver(horizon=H, dtype="f32", algo="group")
        a.set_option(capi.OPT_GROUP_LANES, G)
        a._check(a._lib.tpc_mpc_x_set_group_share(a._h, per_simd * 4 * cus, 0))
        b = MpcSolver(horizon=H, dtype="f32", algo="lane_fma")
        a.reserve(k); b.reserve(n - k)
        f, r = torch.empty_like(sv), torch.empty_like(sv)
        sa, sb = torch.cuda.Stream(dev), torch.cuda.Stream(dev)

        def both():
            with torch.cuda.stream(sa):
                a.solve_batch_compact(sv[:k], sy[:k], sp[:k], out=(f[:k], r[:k]), want_flags=False)
            with torch.cuda.stream(sb):
                b.solve_batch_compact(sv[k:], sy[k:], sp[k:], out=(f[k:], r[k:]), want_flags=False)
        ms = timed(both)
        # each part alone (what bounds the pair from below)
        ma = timed(lambda: a.solve_batch_compact(sv[:k], sy[:k], sp[:k], out=(f[:k], r[:k]), want_flags=False), 5)
        mb = timed(lambda: b.solve_batch_compact(sv[k:], sy[k:], sp[k:], out=(f[k:], r[k:]), want_flags=False), 5)
        # sanity: the LANE_FMA part equals the whole-batch LANE_FMA result bit for bit (same family, another batch)
        same = bool(torch.equal(f[k:], base["lane_fma"][1][order][k:]))
        print(f"top {phi:4.2f} through GROUP G={G} at {per_simd} wavefront(s) per SIMD + rest through LANE_FMA, concurrently: {ms:7.3f} ms "
              f"= {n / ms / 1e3:6.1f} M solves/s   (parts alone: {ma:.3f} + {mb:.3f} ms; LANE_FMA part bit-equal to the whole-batch run: {same})", flush=True)
        a.close(); b.close()
