"""Diagnostic: does the library find the GPU whichever of {libtpc_mpc.so, torch} is loaded first?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
order = sys.argv[1]
if order == "lib_first":
    import trajectory_controller_amd as pkg
    pkg.load_library()
    import torch
    print("torch sees", torch.cuda.is_available())
else:
    import torch
    print("torch sees", torch.cuda.is_available())
    import trajectory_controller_amd as pkg
    pkg.load_library()
from trajectory_controller_amd import MpcSolver
try:
    with MpcSolver(horizon=4, device=0) as s:
        print(order, "solve_one ->", s.solve_one(1.0, 0.1, 0.05))
except Exception as e:
    print(order, "FAILED:", e)
maps = [l.split()[-1] for l in open("/proc/self/maps") if "amdhip" in l or "hsa-runtime" in l]
print(sorted(set(maps)))
