#!/usr/bin/env python3
"""Static count of the instructions a kernel's hottest loop executes per trip when every forward conditional branch
that skips a block IS taken (the rarely-run blocks -- publishing a result, refill -- are jumped over):
    python scripts/isa_hot.py file.s kernel_name_substring [-v]
Walks the largest innermost loop from its header; at a forward s_cbranch inside the loop it follows the branch
(assumes taken), at the backward branch it stops."""
import collections, re, sys
sys.path.insert(0, __import__("os").path.dirname(__file__))
from isa_loop import kernels

def walk(body, verbose=False):
    labels, ins = {}, []
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
    lo, hi = max(inner, key=lambda sp: sp[1] - sp[0])
    pc, out, guard = lo, [], 0
    while pc <= hi and guard < 100000:
        guard += 1
        s = ins[pc]
        out.append(s)
        m = re.match(r"s_cbranch\w*\s+(\.LBB\d+_\d+)|s_branch\s+(\.LBB\d+_\d+)", s)
        if m:
            t = labels.get(m.group(1) or m.group(2), -1)
            if t <= pc and pc == hi:
                break
            if lo <= t <= hi and t > pc:
                pc = t
                continue
        pc += 1
    hist = collections.Counter()
    for s in out:
        op = s.split()[0]
        if op.startswith("v_mov") or op.startswith("v_accvgpr"): hist["mov"] += 1
        elif "f64" in op or "f32" in op: hist["valu_fp"] += 1
        elif op.startswith("v_"): hist["valu_other"] += 1
        elif op.startswith("s_nop"): hist["s_nop"] += 1
        elif op.startswith("s_"): hist["salu"] += 1
        elif op.startswith("ds_"): hist["lds"] += 1
        else: hist["mem"] += 1
    if verbose:
        print("\n".join(out))
    return len(out), dict(hist)

if __name__ == "__main__":
    ks = kernels(sys.argv[1])
    for name, body in ks.items():
        if all(a in name for a in sys.argv[2:] if not a.startswith("-")):
            n, h = walk(body, "-v" in sys.argv)
            print(f"{name}\n  hot path of the largest innermost loop: {n} instructions {h}")
