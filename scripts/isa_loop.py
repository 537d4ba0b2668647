#!/usr/bin/env python3
"""Static look at a kernel's hottest loop in a hipcc `-save-temps` .s file (or llvm-objdump -d text):
instruction histogram of the largest innermost loop body (the label..backward-branch span with the most
instructions that contains no other backward branch).
    python scripts/isa_loop.py file.s kernel_name_substring [...]
    python scripts/isa_loop.py --asm-kernel [trajectory_controller_amd/lib/obj/ub_asm.o]
        the hand-written headline kernel (csrc/mpc_ub_pg_asm.h) as it sits in the shipped object: its labels survive
        as symbols, so the two halves of the loop (LA..LB, LB..SA), the no-stop-test copies of the sweep (NTA*, NTB*)
        and the addresses they start on are read off llvm-objdump -d"""
import collections, re, sys

def kernels(path):
    cur, body, out = None, [], {}
    for line in open(path, errors="ignore"):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur, body = m.group(1), []
            out[cur] = body
            continue
        if cur is not None:
            if line.startswith("\t.end_amdhsa_kernel") or line.startswith(".Lfunc_end"):
                cur = None
                continue
            body.append(line.rstrip("\n"))
    return out

def loops(body):
    labels = {}
    ins = []
    for l in body:
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            labels[m.group(1)] = len(ins)
            continue
        s = l.strip()
        if not s or s.startswith((";", ".", "//")):
            continue
        ins.append(s.split(";")[0].strip())
    spans = []
    for i, s in enumerate(ins):
        m = re.match(r"s_cbranch\w*\s+(\.LBB\d+_\d+)|s_branch\s+(\.LBB\d+_\d+)", s)
        if m:
            t = m.group(1) or m.group(2)
            if t in labels and labels[t] <= i:
                spans.append((labels[t], i))
    inner = [sp for sp in spans if not any(o != sp and sp[0] <= o[0] and o[1] <= sp[1] for o in spans)]
    return ins, inner

def classify(op):
    if op.startswith("v_accvgpr"): return "accvgpr"
    if op.startswith(("ds_",)): return "lds"
    if op.startswith(("scratch_", "buffer_")): return "scratch/buffer"
    if op.startswith(("global_", "flat_")): return "global"
    if op.startswith("s_waitcnt"): return "s_waitcnt"
    if op.startswith("s_nop"): return "s_nop"
    if op.startswith("s_"): return "salu"
    if op.startswith("v_mov"): return "v_mov"
    if re.match(r"v_\w+_f64", op): return "valu_f64"
    if re.match(r"v_\w+_f32", op): return "valu_f32"
    if op.startswith("v_"): return "valu_other"
    return "other"

def asm_kernel(obj):
    import os, subprocess, tempfile
    llvm = "/opt/rocm/lib/llvm/bin/"
    with tempfile.TemporaryDirectory() as td:
        o = os.path.join(td, "k.o")
        open(o, "wb").write(open(obj, "rb").read())
        subprocess.run([llvm + "llvm-objdump", "--offloading", o], cwd=td, capture_output=True)
        co = [f for f in os.listdir(td) if "gfx950" in f]
        assert co, "no gfx950 code object in " + obj
        text = subprocess.run([llvm + "llvm-objdump", "-d", os.path.join(td, co[0])], capture_output=True, text=True).stdout
    blocks, cur = collections.OrderedDict(), None
    for l in text.splitlines():
        m = re.match(r"^([0-9a-f]+) <(\w+)>:", l)
        if m:
            cur = m.group(2); blocks[cur] = (int(m.group(1), 16), [])
            continue
        if cur and l.startswith("\t"):
            blocks[cur][1].append(l.strip().split("//")[0].strip())
    def span(first, until):
        names = list(blocks); a, b = names.index(first), names.index(until)
        return [i for n in names[a:b] for i in blocks[n][1]]
    def show(title, ins, addr):
        h = collections.Counter(classify(i.split()[0]) for i in ins)
        ops = collections.Counter(i.split()[0] for i in ins)
        print(f"{title}: {len(ins)} instructions at {addr:#x} (mod 8 = {addr % 8})  {dict(h)}")
        print("    " + ", ".join(f"{k}:{v}" for k, v in ops.most_common(12)))
    names = list(blocks)
    for sfx in sorted({n[2:] for n in names if re.fullmatch(r"LA\d+", n)}, key=int):   # one asm statement per kernel: its labels end in its number
        kern = [n for n in names[:names.index("LA" + sfx)] if n.startswith("_Z")][-1]
        print(f"== {kern}")
        show("half A (LA..LB)", span("LA" + sfx, "LB" + sfx), blocks["LA" + sfx][0])
        show("half B (LB..SA)", span("LB" + sfx, "SA" + sfx), blocks["LB" + sfx][0])
        nta = [n for n in names if re.fullmatch(r"NTA\d+" + sfx, n) and len(n) > 3 + len(sfx)]
        ntb = [n for n in names if re.fullmatch(r"NTB\d+" + sfx, n) and len(n) > 3 + len(sfx)]
        # (a label NTA17 of statement 0 prints as NTA170: keep those whose step number is a horizon step and whose address follows LA of this statement)
        lo = blocks["SA" + sfx][0]
        nxt = min([blocks[n][0] for n in names if n.startswith("_Z") and blocks[n][0] > lo] + [1 << 62])
        nta = [n for n in nta if lo < blocks[n][0] < nxt]; ntb = [n for n in ntb if lo < blocks[n][0] < nxt]
        if nta and ntb:
            show(f"no-stop-test copy A ({nta[0]}..)", span(nta[0], ntb[0]), blocks[nta[0]][0])
            show(f"no-stop-test copy B ({ntb[0]}..)", span(ntb[0], "XODD" + sfx), blocks[ntb[0]][0])
            loop = span("LA" + sfx, "SA" + sfx)
            print("    entries:", " ".join(f"{n}@{blocks[n][0] % 8}" for n in (nta + ntb) if any(i.split()[-1] == n for i in loop)))

if __name__ == "__main__":
    if sys.argv[1] == "--asm-kernel":
        import os
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        asm_kernel(sys.argv[2] if len(sys.argv) > 2 else os.path.join(root, "trajectory_controller_amd", "lib", "obj", "ub_asm.o"))
        sys.exit(0)
    ks = kernels(sys.argv[1])
    for name, body in ks.items():
        if not all(p in name for p in sys.argv[2:]):
            continue
        ins, inner = loops(body)
        if not inner:
            print(name, "no loop"); continue
        a, b = max(inner, key=lambda sp: sp[1] - sp[0])
        h = collections.Counter(classify(s.split()[0]) for s in ins[a:b + 1])
        ops = collections.Counter(s.split()[0] for s in ins[a:b + 1])
        print(f"{name}\n  hottest innermost loop: {b - a + 1} instructions  {dict(h)}")
        print("  top ops:", ", ".join(f"{k}:{v}" for k, v in ops.most_common(14)))
