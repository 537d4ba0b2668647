"""The shipped library, inspected statically (no GPU): which kernels touch scratch (spill) memory inside an
innermost loop.  Background: one build of the two-variables-per-lane WAVE kernel ran out of
AGPRs, kept a dword of a Hessian entry in scratch and reloaded it inside its iteration loops -- and returned wrong
controls for one instance in nine, while every scratch-free build of the same source agrees with dlib to 1e-13
(caught by test_wave_queue_vs_oracle[40] on the GPU).  The cause was not found (another build that spills
inside its loops is correct, so the spill is a marker of that build, not the bug), so the kernels the BASELINE configs
run -- the compact-form WAVE kernels and the resident single-solve kernels -- must not spill inside a loop; it would
also be a performance bug there.  (The LANE fp64 N = 40 kernels spill in their loop by design and are checked bit for
bit against dlib; general-form WAVE kernels that do are listed, not refused.)"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
LIB = os.path.join(ROOT, "trajectory_controller_amd", "lib", "libtpc_mpc.so")


@pytest.mark.skipif(not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-objdump"), reason="needs llvm-objdump")
def test_no_scratch_inside_wave_loops():
    import __graft_entry__
    if not os.path.exists(LIB):
        __graft_entry__.build()
    import check_loop_scratch
    bad = check_loop_scratch.offenders(LIB, ["wave_kernel", "wave_queue_kernel", "one_shot_kernel"])
    must_be_clean = [name for name, _ in bad if "CompactModel" in name or "one_shot_kernel" in name]
    for name, n in bad:
        print(f"{n} scratch accesses inside an innermost loop: {name}")
    assert not must_be_clean, must_be_clean
