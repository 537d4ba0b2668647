#!/usr/bin/env python3
"""Post-register-allocation rescheduler for the straight-line iteration body of the LANE
projected-gradient kernels (gfx950 assembly as printed by `hipcc -S --cuda-device-only`).

Why: with one wavefront per SIMD an fp64 VALU instruction issues every ~2.0 ns only if it does not
read a result produced in the last few issue slots (scripts/ubench_depdist.hip on MI355X:
distance 1 -> 3.7 ns, 2 -> 2.7, 3 -> 2.35, 4 -> 2.2, 6 -> 2.1, 8 -> 2.05 ns per instruction).  LLVM's
machine model gives fp64 VALU results a latency of one issue slot, so its schedule is full of
distance-2 and distance-3 pairs (about 15 % of the iteration time, matching the SQ wait counters).
This tool re-orders the loop body with a list scheduler that knows the measured penalties.  It never
changes an instruction or a register: only the order, under the full set of register (RAW / WAR /
WAW on 32-bit units), LDS-queue and s_waitcnt dependencies, so results stay bit-identical.

  resched.py in.s out.s [--kernels REGEX] [--report]

Loops handled: basic blocks that are a single-block inner loop (`Inner Loop Header: Depth=2` ...
backward s_cbranch) inside kernels whose name matches --kernels.  Inside such a block only maximal
runs of "plain" instructions (fp64/fp32 VALU arithmetic, moves, v_accvgpr moves, ds_read/ds_write,
s_waitcnt lgkmcnt) are re-ordered; every other instruction is a fence that nothing crosses, so
hazard padding (s_nop) and scalar control code stay exactly where the compiler put them.
"""
import re
import sys

PLAIN_VALU = re.compile(
    r"^(v_(add|mul|max|min|fma|fmac)_f(64|32)(_e32|_e64)?|v_mov_b64(_e32)?|v_mov_b32(_e32)?|"
    r"v_accvgpr_(read|write|mov)_b32|v_(sub|subrev)_f32(_e32|_e64)?|v_pk_(add|mul|fma)_f32)$")
LDS = re.compile(r"^ds_(read|write)")
WAITCNT = re.compile(r"^s_waitcnt$")

# extra issue slots lost when an instruction reads a VALU result produced d instructions earlier
PENALTY = {1: 0.90, 2: 0.35, 3: 0.18, 4: 0.10, 5: 0.06, 6: 0.04, 7: 0.02}
LDS_LATENCY = 28      # issue slots between a ds_read and the s_waitcnt that covers it (~110 cycles)


def reg_units(tok):
    """32-bit register units named in an operand token."""
    out = []
    for m in re.finditer(r"\b([vas])\[(\d+):(\d+)\]|\b([vas])(\d+)\b|\b(vcc|exec|scc|m0)\b", tok):
        if m.group(1):
            out += [f"{m.group(1)}{r}" for r in range(int(m.group(2)), int(m.group(3)) + 1)]
        elif m.group(4):
            out.append(f"{m.group(4)}{m.group(5)}")
        else:
            out.append(m.group(6))
    return out


class Ins:
    __slots__ = ("text", "op", "defs", "uses", "kind", "idx", "lgkm")

    def __init__(self, text, idx):
        self.text, self.idx = text, idx
        body = text.split(";")[0].strip()
        parts = body.split(None, 1)
        self.op = parts[0]
        ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
        self.defs, self.uses, self.lgkm = [], [], None
        if WAITCNT.match(self.op):
            self.kind = "wait"
            m = re.search(r"lgkmcnt\((\d+)\)", body)
            if not m or "vmcnt" in body or "expcnt" in body:
                self.kind = "fence"
            else:
                self.lgkm = int(m.group(1))
        elif LDS.match(self.op):
            self.kind = "lds"
            if self.op.startswith("ds_read"):
                self.defs = reg_units(ops[0])
                for o in ops[1:]:
                    self.uses += reg_units(o.split()[0])
            else:
                for o in ops:
                    self.uses += reg_units(o.split()[0])
        elif PLAIN_VALU.match(self.op):
            self.kind = "valu"
            self.defs = reg_units(ops[0])
            for o in ops[1:]:
                self.uses += reg_units(o)
            if "fmac" in self.op:
                self.uses += self.defs
        else:
            self.kind = "fence"


def schedule_run(run):
    """Re-order one run of plain instructions; returns the new list."""
    n = len(run)
    if n < 8:
        return run
    preds = [dict() for _ in range(n)]      # pred index -> (latency, is_valu_raw)

    def edge(a, b, lat=0, raw=False):
        if a == b:
            return
        old = preds[b].get(a)
        if old is None or lat > old[0] or (raw and not old[1]):
            preds[b][a] = (max(lat, old[0]) if old else lat, raw or (old[1] if old else False))

    last_def, readers = {}, {}
    chain_prev = None
    lds_seen = []                            # indices of LDS ops in order
    pending_read = {}                        # unit -> index of the ds_read that wrote it (not yet waited)
    waits = []
    for i, ins in enumerate(run):
        if ins.kind in ("lds", "wait"):
            if chain_prev is not None:
                edge(chain_prev, i)
            chain_prev = i
        if ins.kind == "wait":
            waits.append(i)
            # which earlier ds ops does this wait cover: all but the last `lgkm` issued
            covered = lds_seen[:len(lds_seen) - ins.lgkm] if ins.lgkm else list(lds_seen)
            for d in covered:
                if run[d].op.startswith("ds_read"):
                    edge(d, i, LDS_LATENCY)
            cov = set(covered)
            for u, d in list(pending_read.items()):
                if d in cov:
                    pending_read[u] = ("done", i)     # later touchers must follow this wait
            continue
        for u in ins.uses + ins.defs:
            pr = pending_read.get(u)
            if pr is not None:
                if isinstance(pr, tuple):
                    edge(pr[1], i)
                else:
                    raise RuntimeError(f"register {u} of an un-waited ds_read touched by: {ins.text.strip()}")
        for u in ins.uses:                   # RAW
            d = last_def.get(u)
            if d is not None:
                edge(d, i, 1, run[d].kind == "valu")
        for u in ins.defs:                   # WAW, WAR
            d = last_def.get(u)
            if d is not None:
                edge(d, i)
            for r in readers.get(u, ()):
                edge(r, i)
        for u in ins.uses:
            readers.setdefault(u, []).append(i)
        for u in ins.defs:
            last_def[u] = i
            readers[u] = []
            pending_read.pop(u, None)
        if ins.kind == "lds":
            lds_seen.append(i)
            if ins.op.startswith("ds_read"):
                for u in ins.defs:
                    pending_read[u] = i

    succs = [[] for _ in range(n)]
    for b in range(n):
        for a in preds[b]:
            succs[a].append(b)
    # critical-path height with a nominal 4-slot latency on VALU RAW edges
    height = [0.0] * n
    for a in range(n - 1, -1, -1):
        h = 0.0
        for b in succs[a]:
            lat, raw = preds[b][a]
            w = 4.0 if raw else max(1.0, float(lat))
            h = max(h, height[b] + w)
        height[a] = h

    npred = [len(p) for p in preds]
    avail = [i for i in range(n) if npred[i] == 0]
    pos = [None] * n
    order = []
    t = 0
    while avail:
        best, best_key = None, None
        for c in avail:
            pen = 0.0
            for a, (lat, raw) in preds[c].items():
                d = t - pos[a]
                if raw:
                    pen = max(pen, PENALTY.get(d, 0.0))
                elif lat > 1 and d < lat:
                    pen = max(pen, 0.05 * (lat - d))
            key = (pen - 0.004 * height[c], run[c].idx)
            if best_key is None or key < best_key:
                best, best_key = c, key
        avail.remove(best)
        pos[best] = t
        order.append(best)
        t += 1
        for b in succs[best]:
            npred[b] -= 1
            if npred[b] == 0:
                avail.append(b)
    assert len(order) == n
    return [run[i] for i in order]


def estimate(run):
    """Issue slots lost to short dependency distances in a given order (same model)."""
    last = {}
    lost = 0.0
    for i, ins in enumerate(run):
        if ins.kind == "valu" or ins.kind == "lds":
            pen = 0.0
            for u in ins.uses:
                d = last.get(u)
                if d is not None:
                    pen = max(pen, PENALTY.get(i - d, 0.0))
            lost += pen
        if ins.kind == "valu":
            for u in ins.defs:
                last[u] = i
        else:
            for u in ins.defs:
                last.pop(u, None)
    return lost


def process_block(lines, report, name):
    """lines: the text lines of one loop body.  Returns the re-ordered lines."""
    out, run, before, after, count = [], [], 0.0, 0.0, 0

    def flush():
        nonlocal run, before, after, count
        if run:
            new = schedule_run(run)
            before += estimate(run)
            after += estimate(new)
            count += len(run)
            out.extend(i.text for i in new)
            run = []

    idx = 0
    for ln in lines:
        s = ln.strip()
        if not s or s.startswith(";") or s.startswith(";;"):
            if s.startswith(";;#ASM"):
                continue                      # empty inline-asm markers (register pins): no code
            if "sched_barrier" in s:
                continue
            out_comment = ln
            # comments stay with the fence structure: flush so they do not float
            flush()
            out.append(out_comment)
            continue
        ins = Ins(ln, idx)
        idx += 1
        if ins.kind == "fence":
            flush()
            out.append(ln)
        else:
            run.append(ins)
    flush()
    if report:
        print(f"resched: {name}: {count} plain instructions, estimated lost issue slots "
              f"{before:.0f} -> {after:.0f}", file=sys.stderr)
    return out


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    opts = [a for a in sys.argv[1:] if a.startswith("--")]
    src, dst = args[0], args[1]
    kre = re.compile(next((o.split("=", 1)[1] for o in opts if o.startswith("--kernels=")), "lane_pg_fused_kernel"))
    report = "--report" in opts
    lines = open(src).read().split("\n")
    out = []
    i, n = 0, len(lines)
    cur_fn = None
    while i < n:
        ln = lines[i]
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            cur_fn = m.group(1)
        if ln.startswith(".Lfunc_end"):
            cur_fn = None
        if cur_fn and kre.search(cur_fn) and "Inner Loop Header: Depth=2" in ln:
            # the label line precedes this comment; find the closing backward branch of this block
            label = None
            for back in range(len(out) - 1, max(len(out) - 4, -1), -1):
                lm = re.match(r"^(\.LBB\d+_\d+):", out[back])
                if lm:
                    label = lm.group(1)
                    break
            j = i + 1
            body = []
            ok = False
            while j < n:
                s = lines[j].strip()
                if re.match(r"^\.LBB\d+_\d+:", s) or s.startswith("; %bb."):
                    break                     # another block starts: not a single-block loop
                if s.startswith("s_cbranch") or s.startswith("s_branch"):
                    ok = label is not None and s.split()[-1] == label
                    break
                body.append(lines[j])
                j += 1
            if ok:
                out.append(ln)
                out.extend(process_block(body, report, cur_fn[:60]))
                i = j
                continue
        out.append(ln)
        i += 1
    open(dst, "w").write("\n".join(out))


if __name__ == "__main__":
    main()
