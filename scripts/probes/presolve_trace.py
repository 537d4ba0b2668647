#!/usr/bin/env python3
"""Kernel timeline of ONE compact 16 384 x N = 40 AUTO solve (AUTO's presolve beside GROUP's pass):
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/pt -- python3 scripts/probes/presolve_trace.py run
    python3 scripts/probes/presolve_trace.py show gpurun_out/pt"""
import csv, glob, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if sys.argv[1] == "run":
    import torch
    from trajectory_controller_amd import MpcSolver
    from trajectory_controller_amd.synth import compact_inputs
    H, n = 40, int(sys.argv[2]) if len(sys.argv) > 2 else 16384
    v, dy, dphi = (torch.from_numpy(a).cuda() for a in compact_inputs(H, n))
    with MpcSolver(horizon=H, algo="auto") as s:
        for _ in range(2):
            s.solve_batch_compact(v, dy, dphi)
            torch.cuda.synchronize()
else:
    path = glob.glob(os.path.join(sys.argv[2], "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
    idx = max(i for i, r in enumerate(rows) if "ub_cd_kernel" in r["Kernel_Name"] or "lane_cd_kernel" in r["Kernel_Name"] and "Li2E" in r["Kernel_Name"][-40:])
    # the last call: from the last SUBSET-2 coordinate-descent kernel on
    starts = [i for i, r in enumerate(rows) if "lane_cd_kernel" in r["Kernel_Name"]]
    i0 = starts[-2] if len(starts) >= 2 else 0
    t0 = int(rows[i0]["Start_Timestamp"])
    for r in rows[max(0, i0 - 2):]:
        name = r["Kernel_Name"].split("(")[0].replace("void tpc::", "").replace("tpc::", "")[:64]
        print(f"{(int(r['Start_Timestamp']) - t0) / 1e6:8.3f} .. {(int(r['End_Timestamp']) - t0) / 1e6:8.3f} ms  q{r.get('Queue_Id', '?'):>3}  grid {r.get('Grid_Size', r.get('Grid_Size_X', '?')):>7}  {name}")
