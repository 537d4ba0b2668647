import sys, numpy as np, torch
sys.path.insert(0,'/root/repo')
from tests.model.bindings import UbModel
from trajectory_controller_amd import MpcSolver
from trajectory_controller_amd.synth import compact_inputs
m=UbModel(); bad=0
for H in (4,5,10,20):
    for n in (1,2,63,64,65,127,129,1000,4097,70001):
        v,dy,dphi=compact_inputs(H,n,first=12345+n)
        mf,mr,mit,_=m.solve_compact(H,v,dy,dphi,nthreads=8)
        with MpcSolver(horizon=H,algo="lane_fma") as s:
            f,r,it=s.solve_batch_compact(*(torch.from_numpy(a).cuda() for a in (v,dy,dphi)),want_iters=True)
        ok=np.array_equal(f.cpu().numpy().view(np.uint64),mf.view(np.uint64)) and np.array_equal(r.cpu().numpy().view(np.uint64),mr.view(np.uint64)) and np.array_equal(it.cpu().numpy(),mit)
        bad+=not ok
        print(H,n,"ok" if ok else "MISMATCH")
print("mismatches",bad)
