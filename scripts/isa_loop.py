#!/usr/bin/env python3
"""Static look at a kernel's hottest loop in a hipcc `-save-temps` .s file (or llvm-objdump -d text):
instruction histogram of the largest innermost loop body (the label..backward-branch span with the most
instructions that contains no other backward branch).
    python scripts/isa_loop.py file.s kernel_name_substring [...]"""
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

if __name__ == "__main__":
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
