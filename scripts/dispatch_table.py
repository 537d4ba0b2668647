#!/usr/bin/env python3
"""AUTO's dispatch table as markdown, from csrc/auto_table.h (the one DESIGN.md section 4 shows):
    python scripts/dispatch_table.py"""
import os, re
src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "trajectory_controller_amd", "csrc", "auto_table.h")).read()
rows = re.findall(r"\{(\d), (\d), (\d+),\s*(\d+),\s*(\d+),\s*(\d+),\s*(\d+),\s*(\d+)\}", src)
NEVER = 1 << 40
fmt = lambda x: "never overtaken" if int(x) >= NEVER else f"{int(x):,}".replace(",", " ")
print("| form | dtype | N | WAVE below | GROUP G=8 below | G=4 below | G=2 below | then | two wavefronts per SIMD from |")
print("|---|---|---|---|---|---|---|---|---|")
for form, dt, H, w, g8, g4, g2, pair in rows:
    one = "LANE_FMA" if (form == "0" or int(H) <= 20) else "LANE"
    print(f"| {'compact' if form == '0' else 'general'} | {'fp64' if dt == '0' else 'fp32'} | {H} | {fmt(w)} | {fmt(g8)} | {fmt(g4)} | {fmt(g2)} | {one} | {fmt(pair) if int(pair) < NEVER else '-'} |")
print("\n(a column equal to the one before it: that family never wins at that horizon; N = 4, 5: WAVE, then LANE_FMA; batch sizes are for a 256-CU part and scale with the CU count)")
