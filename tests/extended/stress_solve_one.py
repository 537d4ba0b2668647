import sys, time; sys.path.insert(0, ".")
import numpy as np, torch
from oracle.bindings import Oracle
from trajectory_controller_amd import MpcSolver
from trajectory_controller_amd.synth import compact_inputs
orc = Oracle()
t0 = time.time()
hs = [4, 5, 10, 20, 30, 4, 10, 4]
solvers = [MpcSolver(horizon=h) for h in hs]
exp = {}
for h in set(hs):
    v, dy, dphi = compact_inputs(h, 64, first=5)
    exp[h] = (v, dy, dphi) + orc.solve_compact(h, v, dy, dphi)[:2]
big = MpcSolver(horizon=20, algo="lane")
bv = [torch.from_numpy(a).cuda() for a in compact_inputs(20, 131072)]
bad = 0
for rep in range(400):
    for s, h in zip(solvers, hs):
        v, dy, dphi, f0, r0 = exp[h]
        k = rep % 64
        hh = h if rep % 37 else (10 if h != 10 else 4)       # now and then another horizon on the same handle: the wave is swapped
        f, r = s.solve_one(v[k], dy[k], dphi[k], horizon=hh)
        if hh == h and (abs(f - f0[k]) > 1e-9 or abs(r - r0[k]) > 1e-9):
            bad += 1
    if rep % 50 == 0:
        big.solve_batch_compact(*bv)            # a batch solve beside the resident waves
        torch.cuda.synchronize()               # a device-wide sync while they are up (waits <= idle timeout)
    if rep % 100 == 99:
        time.sleep(0.03)                       # let them time out, then go on
for s in solvers:
    s.close()
big.close()
print(f"stress: {400 * len(hs)} solve_one calls on {len(hs)} handles, {bad} wrong, {time.time() - t0:.1f} s")
assert bad == 0
