#!/usr/bin/env python3
"""Static audit of hipcc `-S --cuda-device-only` output: kernels whose LDS footprint, not their registers, decides how many
wavefronts a CU holds -- and holds fewer than 8 (160 KB of LDS per CU, 512 registers per SIMD lane).  A kernel that
ends up below FOUR has idle SIMDs (the coordinate-descent kernels did, before the residency rule in csrc/mpc_ub.h).
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Icsrc -Iinclude -DTPC_LANE_H=20 -S --cuda-device-only \\
          -o /tmp/lane20.s trajectory_controller_amd/csrc/mpc_lane_inst.hip && python scripts/lds_residency.py /tmp/lane20.s"""
import re, sys

def parse(path):
    txt = open(path, errors="ignore").read()
    i = txt.find("amdhsa.kernels:")
    for blk in txt[i:].split("  - .agpr_count:")[1:]:
        g = lambda k: int(re.search(r"\." + k + r":\s+(\d+)", blk).group(1))
        yield re.search(r"\.name:\s+(\S+)", blk).group(1), g("group_segment_fixed_size"), g("vgpr_count"), g("max_flat_workgroup_size")

for f in sys.argv[1:]:
    for name, lds, regs, wg in parse(f):
        by_lds = (160 * 1024 // lds) * max(1, wg // 64) if lds else 99
        by_reg = 4 * max(1, 512 // (((regs + 7) // 8) * 8))
        if by_lds < by_reg and by_lds < 8:
            print(f"{re.sub(r'^_ZN3tpc[0-9]+', '', name)[:96]:98s} LDS {lds:6d} B, {regs:3d} registers: {by_lds} wavefronts per CU by LDS, {by_reg} by registers")
