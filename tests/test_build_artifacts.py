"""The shipped library, inspected statically (no GPU): which kernels touch scratch (spill) memory inside an
innermost loop.  Background ("a build that was wrong"): one build of the two-variables-per-lane WAVE
kernel ran out of AGPRs, kept a dword of a Hessian entry in scratch and reloaded it inside its iteration loops --
and returned wrong controls for one instance in nine, while every scratch-free build of the same source agrees with
dlib to 1e-13 (caught by test_wave_queue_vs_oracle[40] on the GPU).  The cause was never pinned to an instruction, so
its precondition is removed instead: NO kernel of the WAVE family -- compact or general form, plain, queue, grouped,
the queue-order kernel, the resident single-solve kernels -- no kernel of the LANE_FMA family (ub_*: since round 5 its
fp64 one-lane kernels at N = 30 / 40, which parked their state in scratch, are no longer built: GROUP takes those requests), no general-model LANE_FMA kernel (ubg_*) and
no GROUP kernel and no G-lanes-per-instance LANE kernel (group_pg_kernel, groupg_pg_kernel, lanex_pg_kernel, lanexg_pg_kernel: built for one wavefront per SIMD with the whole register file) may access scratch
inside a loop.  (The bit-exact LANE kernels at N = 40 -- the state-returning / general long-horizon path -- still
spill in their loops; they are checked bit for bit against dlib on the GPU and are listed.)"""
import re
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
LIB = os.path.join(ROOT, "trajectory_controller_amd", "lib", "libtpc_mpc.so")

MUST_BE_CLEAN = re.compile(r"wave_|one_shot_kernel|ub_pg_|ub_cd_|ubg_|group_pg_kernel|groupg_pg_kernel|lanexg?_pg_kernel")


@pytest.mark.skipif(not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-objdump"), reason="needs llvm-objdump")
def test_no_scratch_inside_wave_loops():
    import __graft_entry__
    if not os.path.exists(LIB):
        __graft_entry__.build()
    import check_loop_scratch
    bad = check_loop_scratch.offenders(LIB, [])
    refused = [name for name, _ in bad if MUST_BE_CLEAN.search(name)]
    for name, n in bad:
        print(f"{n} scratch accesses inside an innermost loop: {name}")
    assert not refused, refused
    # the listed rest is exactly the bit-exact LANE kernels at N = 40 known to spill (a new name here wants a look)
    assert all("Li40E" in name and "lane_" in name for name, _ in bad), [name for name, _ in bad]


def test_auto_table_matches_its_records():
    """csrc/auto_table.h is GENERATED from the crossover records committed under profiles/ (scripts/measure_crossover.py
    --from-record): re-derive the rows here, without a GPU, and hold the header to them -- a hand edit of the table, or a
    record replaced without regenerating it, fails."""
    import subprocess
    hdr = open(os.path.join(ROOT, "trajectory_controller_amd", "csrc", "auto_table.h")).read()
    rows = {}
    for line in hdr.splitlines():
        line = line.strip()
        if line.startswith("{") and line.endswith("},"):
            nums = [int(x) for x in line.strip("{},").split(",")]
            rows[(nums[0], nums[1], nums[2])] = tuple(nums[3:])
    assert len(rows) == 12 and all(len(r) == 5 for r in rows.values())
    never = 1 << 40
    derived = {}
    for form, records in ((0, "profiles/r04_crossover.txt,profiles/r04_crossover_top.txt,profiles/r04_crossover_f32.txt,profiles/r04_crossover_f32_top.txt,profiles/r05_crossover_h10.txt"),
                          (1, "profiles/r04_crossover_general.txt,profiles/r04_crossover_general_top.txt")):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "measure_crossover.py"), "--from-record", records],
                             cwd=ROOT, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr[-2000:]
        for m in re.finditer(r"=> (f64|f32) H=(\d+): WAVE (never overtaken|below \d+), GROUP 8 (never overtaken|below \d+), 4 (never overtaken|below \d+), "
                             r"2 (never overtaken|below \d+), LANE_FMA from there(?:; GROUP grid two wavefronts per SIMD from (\d+))?", out.stdout):
            val = lambda w: never if w.startswith("never") else int(w.split()[1])
            derived[(form, 0 if m.group(1) == "f64" else 1, int(m.group(2)))] = (
                val(m.group(3)), val(m.group(4)), val(m.group(5)), val(m.group(6)), int(m.group(7)) if m.group(7) else never)
    assert derived == rows, {k: (rows.get(k), derived.get(k)) for k in set(rows) | set(derived) if rows.get(k) != derived.get(k)}


@pytest.mark.parametrize("header, fmas", [("mpc_ub_pg_asm.h", 100), ("mpc_ub_pg_asm_h10.h", 50)])
def test_hand_written_kernel_header_is_what_its_generator_writes(header, fmas):
    """csrc/mpc_ub_pg_asm.h (N = 20) and mpc_ub_pg_asm_h10.h are GENERATED (scripts/gen_ub_pg_asm.py + scripts/ubasm.py, `make -C csrc regen`): the committed
    header must be the generator's output for the shipped arguments -- a hand edit of either side alone fails -- and the
    stream it holds must keep the properties the kernel's speed rests on: every instruction of the loop 8 bytes long except
    an even number of 4-byte scalar ones per half (an 8-byte instruction starting on an odd dword costs a fifth cycle), no
    register copy in the loop, and the no-stop-test copies of the sweep free of stop-test instructions."""
    import subprocess
    gen_args = {"mpc_ub_pg_asm.h": ["3"], "mpc_ub_pg_asm_h10.h": ["0", "0", "0", "0", "2,5", "10"]}[header]   # (csrc/Makefile, regen)
    gen = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "gen_ub_pg_asm.py")] + gen_args, capture_output=True, text=True, timeout=120)
    assert gen.returncode == 0, gen.stderr[-2000:]
    committed = open(os.path.join(ROOT, "trajectory_controller_amd", "csrc", header)).read()
    assert gen.stdout == committed
    lines = [l.strip().strip('"').replace("\\n", "") for l in committed.splitlines() if l.strip().startswith('"')]
    la, lb, sa = lines.index("LA%=:"), lines.index("LB%=:"), lines.index("SA%=:")
    four_byte = ("s_and_b64", "s_andn2_b64", "s_mov_b64", "s_sub_u32", "s_cbranch", "s_branch", "s_add_u32", "s_or_b64", "s_cmp")
    def walk(block):
        """byte offsets of a block that starts on an 8-byte boundary: every 8-byte instruction on an even dword; returns its size"""
        off = 0
        for l in block:
            if l.endswith(":") or l.startswith("."): continue
            if l.startswith(four_byte): off += 4
            else:
                assert l.startswith("v_") and l.split()[0].endswith(("_f64", "_e64", "_b32")), l   # (VOP3 / accvgpr: 8 bytes)
                assert off % 8 == 0, (off, l)
                off += 8
        return off
    half_a, half_b = lines[la + 1:lb], lines[lb + 1:sa]
    assert lines[la - 1] == ".p2align 3"
    assert walk(half_a) % 8 == 0   # (LB follows half A directly)
    walk(half_b)
    for half in (half_a, half_b):
        assert not any(l.startswith(("v_mov_b64", "v_mov_b32")) for l in half)
    nta, ntb = lines.index(next(l for l in lines if l.startswith("NTA") and l.endswith("%=:"))), lines.index(next(l for l in lines if l.startswith("NTB") and l.endswith("%=:")))
    xodd = lines.index("XODD%=:")
    for first, copy in ((nta, lines[nta:ntb]), (ntb, lines[ntb:xodd])):
        assert lines[first - 1] == ".p2align 3"
        walk(copy)
        assert not any(l.startswith(("v_min_f64", "v_max_f64", "v_cmp")) for l in copy)
        assert sum(l.startswith("v_fma_f64") for l in copy) > fmas
