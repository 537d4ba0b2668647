#!/usr/bin/env python3
"""Wave-level reading of tests/extended/stop_depth_stats.cpp (ANALYSIS TOOL, CPU only): every 16th instance of the headline
batch (262 144 x N=20, dlib's defaults) through the LANE_FMA arithmetic, recording after how many steps of the backward
sweep each PG iteration's stop test is decided; then 16 wavefronts x 64 lanes are fed from the longest-first queue in
lockstep (queue order by lambda, as shipped, and by the true iteration count) and the distribution of max-over-lanes depth per wave iteration is printed -- the fraction of a wavefront's iterations
in which the rest of the sweep can run without the stop test (scripts/gen_ub_pg_asm.py CHECKS).

    g++ -O2 -mfma -ffp-contract=off -std=c++17 -o /tmp/stop_depth_stats tests/extended/stop_depth_stats.cpp
    python tests/extended/stop_depth_sim.py [/tmp/stop_depth_stats]
"""
import math, os, subprocess, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from trajectory_controller_amd.synth import compact_inputs

tool = sys.argv[1] if len(sys.argv) > 1 else "/tmp/stop_depth_stats"
H, N, EVERY, WAVES = 20, 262144, 16, 16
v, dy, dphi = compact_inputs(H, N)
idx = np.arange(0, N, EVERY)
np.concatenate([v[idx], dy[idx], dphi[idx]]).tofile("/tmp/sd_in.bin")
am = 22 * math.pi / 180
subprocess.check_call([tool, "/tmp/sd_in.bin", str(len(idx)), "/tmp/sd_out.bin", "20", "7", "0.0005", "10", "0.1", "0.21", repr(-am), repr(am)])
b = open("/tmp/sd_out.bin", "rb").read()
pos, seqs, hist = 0, [], np.zeros(H + 2, dtype=np.int64)
while pos < len(b):
    cnt, _cd = np.frombuffer(b, dtype=np.uint32, count=2, offset=pos); pos += 8
    d = np.frombuffer(b, dtype=np.uint8, count=cnt, offset=pos); pos += cnt
    seqs.append(d)
    hist += np.bincount(d, minlength=H + 2)
c = np.cumsum(hist) / hist.sum()
print("per instance: fraction of PG iterations decided after k steps:", " ".join(f"{k}:{c[k]:.3f}" for k in (1, 2, 3, 5, 8, 12, 20)))
vs = v[idx]
lengths = np.array([len(s) for s in seqs])


def simulate(queue):
    """16 wavefronts x 64 lanes in lockstep, each lane taking the queue's next instance when its own is done"""
    lanes = [[None, 0] for _ in range(WAVES * 64)]
    qi = 0
    for l in range(len(lanes)):
        lanes[l][0], lanes[l][1] = seqs[queue[qi]], 0; qi += 1
    wh = np.zeros(H + 2, dtype=np.int64)
    active = True
    while active:
        active = False
        for w in range(WAVES):
            md = 0
            for l in range(w * 64, (w + 1) * 64):
                s, p = lanes[l]
                if s is None: continue
                md = max(md, int(s[p]))
                if p + 1 < len(s): lanes[l][1] = p + 1
                elif qi < len(queue): lanes[l][0], lanes[l][1] = seqs[queue[qi]], 0; qi += 1
                else: lanes[l][0] = None
            if md:
                wh[md] += 1; active = True
    return np.cumsum(wh) / wh.sum()


# the kernels' queue is longest-first by lambda, a function of the speed (correlation of sqrt(lambda) with the iteration
# count: 0.986, median error of the best fit 9 %); the true-length order is what a perfect predictor would give
for name, key in (("queue by lambda (as shipped)", -vs), ("queue by true iteration count", -lengths)):
    c = simulate([q for q in np.argsort(key, kind="stable") if lengths[q]])
    print(f"{name}: all 64 lanes decided after k steps:", " ".join(f"{k}:{c[k]:.3f}" for k in (1, 2, 3, 4, 5, 8, 10, 12, 16, 19, 20)))
    for ks in ((2,), (3,), (2, 5), (3, 8), (1, 2, 4)):
        saved, prev, cost, reach = 0.0, 0.0, 0.0, 1.0
        for k in ks:
            saved += (c[k] - prev) * 6 * (H - k); cost += 3 * reach; reach = 1 - c[k]; prev = c[k]
        print(f"    checks {ks}: {saved - cost:.1f} of 530 instructions per iteration saved")
