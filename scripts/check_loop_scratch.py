#!/usr/bin/env python3
"""Static check of the built library: which kernels access scratch (spill) memory inside an innermost loop.
usage: check_loop_scratch.py [libtpc_mpc.so] [name-substring ...]     (prints one line per offending kernel; exit 1 if any)

Why: a build of the two-variables-per-lane WAVE kernel whose register allocation ran out of AGPRs kept one dword of a
Hessian entry in scratch and reloaded it inside the iteration loops; that build returned wrong controls for one
instance in nine, the scratch-free builds of the same source do not.  The LANE fp64 H = 40
kernel spills inside its loop by design and is bit-exact, so the rule is applied to the WAVE family only."""
import os, re, shutil, subprocess, sys, tempfile

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def device_objects(lib, tmp):
    dst = os.path.join(tmp, os.path.basename(lib))
    shutil.copy(lib, dst)
    subprocess.run([OBJDUMP, "--offloading", dst], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return sorted(os.path.join(tmp, f) for f in os.listdir(tmp) if "amdgcn" in f)


def kernels(code_object):
    """yield (symbol, [(address, text), ...]) per function of a code object"""
    out = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", code_object], check=True, capture_output=True, text=True).stdout
    name, body = None, []
    for line in out.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m:
            if name:
                yield name, body
            name, body = m.group(1), []
            continue
        m = re.match(r"^\s*([a-z_0-9]+.*?)\s*//\s*([0-9A-Fa-f]+):(.*)$", line)
        if m and name:
            t = re.search(r"<[^>]*?(?:\+0x([0-9a-f]+))?>\s*$", m.group(3))
            off = None if not t else (int(t.group(1), 16) if t.group(1) else 0)
            body.append((int(m.group(2), 16), m.group(1), off))
    if name:
        yield name, body


def loops(body):
    """innermost loops as (first address, last address): backward branches whose range holds no other one"""
    ranges = []
    for addr, text, off in body:
        if text.startswith(("s_cbranch", "s_branch")) and off is not None:
            tgt = body[0][0] + off
            if tgt <= addr:
                ranges.append((tgt, addr))
    return [r for r in ranges if not any(o != r and r[0] <= o[0] and o[1] <= r[1] for o in ranges)]


def offenders(lib, substrings):
    bad = []
    with tempfile.TemporaryDirectory() as tmp:
        for co in device_objects(lib, tmp):
            for name, body in kernels(co):
                if substrings and not any(s in name for s in substrings):
                    continue
                inner = loops(body)
                hits = [(a, t) for a, t, _ in body if t.startswith("scratch_") and any(lo <= a <= hi for lo, hi in inner)]
                if hits:
                    bad.append((name, len(hits)))
    return bad


if __name__ == "__main__":
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, "trajectory_controller_amd", "lib", "libtpc_mpc.so")
    bad = offenders(lib, sys.argv[2:])
    for name, n in bad:
        print(f"{n:4d} scratch accesses inside an innermost loop: {name}")
    sys.exit(1 if bad else 0)
