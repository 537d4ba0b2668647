#!/usr/bin/env python3
"""Generates and (on a GPU box) runs a microbenchmark of instruction-issue cost for ONE wavefront per SIMD on gfx950:
which fp64 forms are cheap (in-place / fresh destination / dependent), what register moves, AGPR transfers, scalar
instructions and LDS accesses ADD to a stream of fp64 arithmetic (hidden or not), register-bank effects.
    python scripts/probes/issue_forms.py gen  > /tmp/issue_forms.hip   (any box)
    hipcc --offload-arch=gfx950 -O3 /tmp/issue_forms.hip -o /tmp/issue_forms && /tmp/issue_forms
Every kernel is one asm statement with explicit registers: a loop of ITER repetitions of a generated body."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ubasm

def R(i): return f"v[{i}:{i+1}]"

def body_single(fmt, n=256, ring=8, stride=2):
    out = []
    for k in range(n):
        d = (k % ring) * stride
        dn = ((k + 1) % ring) * stride
        out.append(fmt.format(d=R(d), dn=R(dn)))
    return out

def fma_stream(n, ring=8):
    return [f"v_fma_f64 {R((k % ring) * 2)}, v[64:65], v[66:67], {R((k % ring) * 2)}" for k in range(n)]

def mix(n_f, every, extra):
    out, e = [], 0
    for k, ins in enumerate(fma_stream(n_f)):
        out.append(ins)
        if (k + 1) % every == 0:
            x = extra(e); e += 1
            out.extend(x if isinstance(x, list) else [x])
    return out

K = {}
# ---- group A: single forms
K["a1 fma acc in place, 1 chain"] = body_single("v_fma_f64 {d}, v[64:65], v[66:67], {d}", ring=1)
K["a2 fma acc in place, 8 chains"] = body_single("v_fma_f64 {d}, v[64:65], v[66:67], {d}")
K["a3 fma fresh dest, independent (8 dests)"] = body_single("v_fma_f64 {d}, v[64:65], v[66:67], v[68:69]")
K["a4 fma fresh dest, src0 = previous result"] = body_single("v_fma_f64 {dn}, {d}, v[66:67], v[68:69]")
K["a4b fma fresh dest, src2 = previous result"] = body_single("v_fma_f64 {dn}, v[64:65], v[66:67], {d}")
K["a5 fma in place through src0"] = body_single("v_fma_f64 {d}, {d}, v[66:67], v[68:69]")
K["a6 fma in place through src1"] = body_single("v_fma_f64 {d}, v[64:65], {d}, v[68:69]")
K["a7 fma acc in place, src0 SGPR"] = body_single("v_fma_f64 {d}, s[20:21], v[66:67], {d}")
K["a7b fma in place src1, src0 SGPR, src2 vgpr"] = body_single("v_fma_f64 {d}, s[20:21], {d}, v[68:69]")
K["a7c fma fresh, src0 SGPR"] = body_single("v_fma_f64 {d}, s[20:21], v[66:67], v[68:69]")
K["a8 add in place"] = body_single("v_add_f64 {d}, {d}, v[64:65]")
K["a9 add fresh independent"] = body_single("v_add_f64 {d}, v[64:65], v[66:67]")
K["a9b add fresh, src = previous result"] = body_single("v_add_f64 {dn}, {d}, v[66:67]")
K["a10 mov_b64 fresh independent"] = body_single("v_mov_b64 {d}, v[64:65]")
K["a11 fma fresh dest, 32 dests"] = body_single("v_fma_f64 {d}, v[64:65], v[66:67], v[68:69]", ring=32)
K["a12 fma acc in place clamp"] = body_single("v_fma_f64 {d}, v[64:65], v[66:67], {d} clamp")
K["a13 fma acc in place neg src"] = body_single("v_fma_f64 {d}, -v[64:65], v[66:67], {d}")
K["a14 min abs abs in place"] = body_single("v_min_f64 {d}, |{d}|, |v[64:65]|")
K["a15 max in place"] = body_single("v_max_f64 {d}, {d}, v[64:65]")
K["a16 fmac VOP2 form"] = body_single("v_fmac_f64 {d}, v[64:65], v[66:67]")
K["a17 mul in place"] = body_single("v_mul_f64 {d}, {d}, v[66:67]")
# ---- group C: banks (register index mod 4 of d / a / b)
for dn_, an, bn in [(0, 64, 68), (0, 66, 68), (0, 66, 70), (2, 64, 68), (2, 66, 70), (0, 64, 64)]:
    K[f"c fma acc in place 1 chain d=v{dn_} a=v{an} b=v{bn}"] = [f"v_fma_f64 {R(dn_)}, {R(an)}, {R(bn)}, {R(dn_)}"] * 256
for dn_, an, bn, cn in [(0, 64, 68, 72), (0, 66, 68, 72), (0, 66, 70, 72), (2, 64, 68, 72), (2, 66, 70, 74), (0, 65 + 1, 68, 74)]:
    K[f"c fma fresh d=v{dn_} a=v{an} b=v{bn} c=v{cn} (same dest every time)"] = [f"v_fma_f64 {R(dn_)}, {R(an)}, {R(bn)}, {R(cn)}"] * 256
# ---- group B: a stream of 384 in-place FMAs plus extras every 4th instruction (96 extras)
K["b0 384 fma baseline"] = fma_stream(384)
K["b1 384 fma + 96 v_mov_b64"] = mix(384, 4, lambda e: f"v_mov_b64 {R(32 + (e % 8) * 2)}, v[70:71]")
K["b2 384 fma + 96 accvgpr (w/r alternating)"] = mix(384, 4, lambda e: (f"v_accvgpr_write_b32 a{e % 16}, v72" if e % 2 == 0 else f"v_accvgpr_read_b32 v{80 + e % 16}, a{(e - 1) % 16}"))
K["b3 384 fma + 96 s_nop 0"] = mix(384, 4, lambda e: "s_nop 0")
K["b4 384 fma + 96 s_mov_b32"] = mix(384, 4, lambda e: f"s_mov_b32 s{30 + e % 4}, s29")
K["b5 384 fma + 24 ds_read_b128 (+wait at end)"] = mix(384, 16, lambda e: f"ds_read_b128 v[{96 + (e % 4) * 4}:{99 + (e % 4) * 4}], v124 offset:{(e % 8) * 1024}") + ["s_waitcnt lgkmcnt(0)"]
K["b6 384 fma + 24 ds_write_b128"] = mix(384, 16, lambda e: f"ds_write_b128 v124, v[96:99] offset:{(e % 8) * 1024}") + ["s_waitcnt lgkmcnt(0)"]
K["b7 384 fma + 24 ds_read2st64_b64 + 24 waits"] = mix(384, 16, lambda e: [f"ds_read2st64_b64 v[{96 + (e % 4) * 4}:{99 + (e % 4) * 4}], v125 offset0:{(e % 8) * 2} offset1:{(e % 8) * 2 + 1}"]) + ["s_waitcnt lgkmcnt(0)"]
K["b8 384 fma + 24 ds_write2st64_b64"] = mix(384, 16, lambda e: f"ds_write2st64_b64 v125, v[96:97], v[100:101] offset0:{(e % 8) * 2} offset1:{(e % 8) * 2 + 1}") + ["s_waitcnt lgkmcnt(0)"]
K["b9 384 fma + 24 (read2st64 .. wait .. use) prefetched one group ahead"] = mix(384, 16, lambda e: [f"s_waitcnt lgkmcnt(0)", f"v_add_f64 v[110:111], v[{96 + ((e + 1) % 2) * 4}:{97 + ((e + 1) % 2) * 4}], v[110:111]", f"ds_read2st64_b64 v[{96 + (e % 2) * 4}:{99 + (e % 2) * 4}], v125 offset0:{(e % 8) * 2} offset1:{(e % 8) * 2 + 1}"]) + ["s_waitcnt lgkmcnt(0)"]
K["b10 384 fma + 96 v_cmp_lt_f64"] = mix(384, 4, lambda e: "v_cmp_lt_f64 vcc, v[70:71], v[72:73]")
K["b11 384 fma + 96 v_add_u32"] = mix(384, 4, lambda e: f"v_add_u32 v{80 + e % 8}, v{80 + e % 8}, v79")
K["b12 384 fma + 96 s_waitcnt lgkmcnt(0) (nothing outstanding)"] = mix(384, 4, lambda e: "s_waitcnt lgkmcnt(0)")
# one step of the planned in-place backward sweep (register-resident step), 25 instructions, x 16 steps
def pg_step(i):
    X0, X1, V0, V1, WZ, WY = (R(8 + 12 * (i % 8) + 2 * j) for j in range(6))
    n0, n1, acc = R(112), R(114), R(116)
    a, c, as_, cs, q1th, il, beta, grs0, grs1 = (R(64 + 2 * j) for j in range(9))
    return [
        f"v_fma_f64 {WZ}, {a}, {WY}, {WZ}", f"v_fma_f64 {WZ}, {as_}, {X1}, {WZ}",
        f"v_fma_f64 {WY}, -{cs}, {X1}, {WY}", f"v_fma_f64 {WY}, {cs}, {X0}, {WY}",
        f"v_fma_f64 {WY}, s[20:21], {WY}, -{q1th}", f"v_fma_f64 {n1}, {a}, {n0}, {n1}", f"v_add_f64 {n1}, {n1}, {WY}",
        f"v_fma_f64 {n0}, s[22:23], {WZ}, {n0}",
        f"v_fma_f64 {WY}, {grs0}, {X0}, s[24:25]", f"v_fma_f64 {WY}, {c}, {n1}, {WY}",
        f"v_fma_f64 {WZ}, -{il}, {WY}, {X0} clamp", f"v_add_f64 {X0}, {X0}, -{WZ}", f"v_min_f64 {X0}, |{WY}|, |{X0}|",
        f"v_max_f64 {acc}, {acc}, {X0}", f"v_add_f64 {V0}, {WZ}, -{V0}", f"v_fma_f64 {X0}, {beta}, {V0}, {WZ} clamp",
        f"v_fma_f64 {V0}, {grs1}, {X1}, s[26:27]", f"v_fma_f64 {V0}, -{c}, {n1}, {V0}", f"v_fma_f64 {V0}, {a}, {n0}, {V0}",
        f"v_fma_f64 {WY}, -{il}, {V0}, {X1} clamp", f"v_add_f64 {X1}, {X1}, -{WY}", f"v_min_f64 {X1}, |{V0}|, |{X1}|",
        f"v_max_f64 {acc}, {acc}, {X1}", f"v_add_f64 {V1}, {WY}, -{V1}", f"v_fma_f64 {X1}, {beta}, {V1}, {WY} clamp",
    ]
K["p1 planned in-place PG step x16 (400 instr)"] = [ins for i in range(16) for ins in pg_step(i)]

K["p2 planned PG step x48 (1200 instr, 9.6 KB of code)"] = [ins for r in range(3) for i in range(16) for ins in pg_step(i)]
K["p3 planned PG step x96 (2400 instr, 19 KB of code)"] = [ins for r in range(6) for i in range(16) for ins in pg_step(i)]
K["b13 1536 fma (12 KB of code)"] = fma_stream(1536)
K["b14 768 v_fmac_f64_e32 VOP2 (4-byte encodings)"] = [f"v_fmac_f64_e32 {R((k % 8) * 2)}, v[64:65], v[66:67]" for k in range(768)]
K["b15 768 fma VOP3 (8-byte)"] = fma_stream(768)

# ---- the shipped iteration loop itself (scripts/ubasm.py), every register hard-coded: two iterations (halves A, B) per body
BIG = set()
OPTS = {}   # name -> (VGPRs clobbered, AGPRs clobbered)
def real_loop(NA, order="plain", control=True):
    P = ubasm.Plan(H=20, NA=NA, all_hard=True, order=order)
    P.S = {"sgq0": "s[20:21]", "sgq1": "s[22:23]", "sgrs0": "s[24:25]", "sgrs1": "s[26:27]", "slo1": "s[20:21]", "sgeps": "s[22:23]",
           "shave": "s[12:13]", "sexec": "s[14:15]", "sleft": "s16"}
    P.A = {k: f"a{j}" for j, k in enumerate(sorted(P.A))}
    tail = ["s_cbranch_vccnz 2f", "s_cbranch_scc1 2f"] if control else []
    body = P.iteration("A", tail) + P.iteration("B", tail)
    if not control:
        body = [l for l in body if not l.startswith(("s_", "v_cmp"))]
    return body
for na in (4, 6):
    nm = f"r{na} shipped loop, two iterations, NA={na}"
    K[nm] = real_loop(na); BIG.add(nm)
nm = "ra4 shipped loop NA=4, loop aligned to 8 bytes"
K[nm] = real_loop(4); BIG.add(nm)
nm = "ra4m shipped loop NA=4, loop mis-aligned (aligned + one s_nop)"
K[nm] = real_loop(4); BIG.add(nm)
K["ya 768 fma, loop aligned to 8 bytes"] = fma_stream(768)
K["ym 768 fma, loop mis-aligned by 4 bytes (aligned + s_nop)"] = fma_stream(768)
nm = "r4n shipped loop NA=4 without its scalar / mask instructions"
K[nm] = real_loop(4, control=False); BIG.add(nm)
def parts(NA, what, copies, strip_agpr=False):
    P = ubasm.Plan(H=20, NA=NA, all_hard=True)
    P.S = {"sgq0": "s[20:21]", "sgq1": "s[22:23]", "sgrs0": "s[24:25]", "sgrs1": "s[26:27]", "slo1": "s[20:21]", "sgeps": "s[22:23]",
           "shave": "s[12:13]", "sexec": "s[14:15]", "sleft": "s16"}
    P.A = {k: f"a{j}" for j, k in enumerate(sorted(P.A))}
    body = []
    for c in range(copies):
        half = "AB"[c % 2]
        body += P.forward(half) if what == "fwd" else [l for l in P.backward(half, []) if not l.startswith(("s_", "v_cmp"))]
    if strip_agpr:
        body = [l for l in body if "accvgpr" not in l]
    return body
for nm, b in (("f forward pass only x6", parts(4, "fwd", 6)), ("g backward sweep only x2 (no control)", parts(4, "bwd", 2)),
              ("g0 backward sweep only x2, AGPR transfers deleted", parts(4, "bwd", 2, True))):
    K[nm] = b; BIG.add(nm)
# the cost of branches: 384 fma in chunks of 32, each chunk ending in ...
def chunks(kind):
    out = []
    for c in range(12):
        out += fma_stream(32)
        if kind == "taken": out += [f"s_branch 9{c}f", "s_nop 0", f"9{c}:"]
        if kind == "cond_not_taken": out += ["s_cmp_eq_u32 s29, 0", "s_cbranch_scc1 2f"]
        if kind == "vccnz_not_taken": out += ["s_cbranch_vccnz 2f"]
        if kind == "exec_toggle": out += ["s_mov_b64 exec, s[14:15]"]
        if kind == "vcmp_sand": out += ["v_cmp_gt_f64 vcc, s[22:23], v[0:1]", "s_and_b64 vcc, vcc, s[12:13]"]
    return out
K["t1 384 fma, 12 taken s_branch (+12 skipped s_nop)"] = chunks("taken")
K["t2 384 fma, 12 x (s_cmp + s_cbranch_scc1 not taken)"] = chunks("cond_not_taken")
K["t3 384 fma, 12 s_cbranch_vccnz not taken"] = chunks("vccnz_not_taken")
K["t4 384 fma, 12 s_mov_b64 exec"] = chunks("exec_toggle")
K["t5 384 fma, 12 x (v_cmp -> s_and vcc)"] = chunks("vcmp_sand")
# p1's arithmetic with distinct registers per step (20 steps x 12 registers) instead of 8 rotating slots
def pg_step_wide(i):
    X0, X1, V0, V1, WZ, WY = (R(12 * i + 2 * j) for j in range(6))
    n0, n1, acc = R(240), R(242), R(244)
    a, c, as_, cs, q1th, il, beta, grs0, grs1 = (R(240 + 2 * (j % 6)) for j in range(9))
    return [l for l in pg_step(0)] and [
        f"v_fma_f64 {WZ}, {a}, {WY}, {WZ}", f"v_fma_f64 {WZ}, {as_}, {X1}, {WZ}",
        f"v_fma_f64 {WY}, -{cs}, {X1}, {WY}", f"v_fma_f64 {WY}, {cs}, {X0}, {WY}",
        f"v_fma_f64 {WY}, s[20:21], {WY}, -{q1th}", f"v_fma_f64 {n1}, {a}, {n0}, {n1}", f"v_add_f64 {n1}, {n1}, {WY}",
        f"v_fma_f64 {n0}, s[22:23], {WZ}, {n0}",
        f"v_fma_f64 {WY}, {grs0}, {X0}, s[24:25]", f"v_fma_f64 {WY}, {c}, {n1}, {WY}",
        f"v_fma_f64 {WZ}, -{il}, {WY}, {X0} clamp", f"v_add_f64 {X0}, {X0}, -{WZ}", f"v_min_f64 {X0}, |{WY}|, |{X0}|",
        f"v_max_f64 {acc}, {acc}, {X0}", f"v_add_f64 {V0}, {WZ}, -{V0}", f"v_fma_f64 {X0}, {beta}, {V0}, {WZ} clamp",
        f"v_fma_f64 {V0}, {grs1}, {X1}, s[26:27]", f"v_fma_f64 {V0}, -{c}, {n1}, {V0}", f"v_fma_f64 {V0}, {a}, {n0}, {V0}",
        f"v_fma_f64 {WY}, -{il}, {V0}, {X1} clamp", f"v_add_f64 {X1}, {X1}, -{WY}", f"v_min_f64 {X1}, |{V0}|, |{X1}|",
        f"v_max_f64 {acc}, {acc}, {X1}", f"v_add_f64 {V1}, {WY}, -{V1}", f"v_fma_f64 {X1}, {beta}, {V1}, {WY} clamp",
    ]
def pg_step_at(i, nslots, base, cbase):
    X0, X1, V0, V1, WZ, WY = (R(base + 12 * (i % nslots) + 2 * j) for j in range(6))
    n0, n1, acc = R(cbase + 18), R(cbase + 20), R(cbase + 22)
    a, c, as_, cs, q1th, il, beta, grs0, grs1 = (R(cbase + 2 * j) for j in range(9))
    return [
        f"v_fma_f64 {WZ}, {a}, {WY}, {WZ}", f"v_fma_f64 {WZ}, {as_}, {X1}, {WZ}",
        f"v_fma_f64 {WY}, -{cs}, {X1}, {WY}", f"v_fma_f64 {WY}, {cs}, {X0}, {WY}",
        f"v_fma_f64 {WY}, s[20:21], {WY}, -{q1th}", f"v_fma_f64 {n1}, {a}, {n0}, {n1}", f"v_add_f64 {n1}, {n1}, {WY}",
        f"v_fma_f64 {n0}, s[22:23], {WZ}, {n0}",
        f"v_fma_f64 {WY}, {grs0}, {X0}, s[24:25]", f"v_fma_f64 {WY}, {c}, {n1}, {WY}",
        f"v_fma_f64 {WZ}, -{il}, {WY}, {X0} clamp", f"v_add_f64 {X0}, {X0}, -{WZ}", f"v_min_f64 {X0}, |{WY}|, |{X0}|",
        f"v_max_f64 {acc}, {acc}, {X0}", f"v_add_f64 {V0}, {WZ}, -{V0}", f"v_fma_f64 {X0}, {beta}, {V0}, {WZ} clamp",
        f"v_fma_f64 {V0}, {grs1}, {X1}, s[26:27]", f"v_fma_f64 {V0}, -{c}, {n1}, {V0}", f"v_fma_f64 {V0}, {a}, {n0}, {V0}",
        f"v_fma_f64 {WY}, -{il}, {V0}, {X1} clamp", f"v_add_f64 {X1}, {X1}, -{WY}", f"v_min_f64 {X1}, |{V0}|, |{X1}|",
        f"v_max_f64 {acc}, {acc}, {X1}", f"v_add_f64 {V1}, {WY}, -{V1}", f"v_fma_f64 {X1}, {beta}, {V1}, {WY} clamp",
    ]
for nm, ns, base, cbase in (("u4 PG step, 8 slots at v0, constants v96", 8, 0, 96), ("u5 PG step, 8 slots at v128, constants v224", 8, 128, 224),
                            ("u6 PG step, 16 slots at v0, constants v192", 16, 0, 192), ("u7 PG step, 19 slots at v0, constants v228", 19, 0, 228),
                            ("u8 PG step, 4 slots at v0, constants v96", 4, 0, 96), ("u9 PG step, 12 slots at v0, constants v192", 12, 0, 192),
                            ("u10 PG step, 2 slots at v0, constants v96", 2, 0, 96)):
    K[nm] = [ins for i in range(48) for ins in pg_step_at(i, ns, base, cbase)]; BIG.add(nm)
def acc_at(dbase, sbase, ring=8):
    return [f"v_fma_f64 {R(dbase + (k % ring) * 2)}, {R(sbase)}, {R(sbase + 2)}, {R(dbase + (k % ring) * 2)}" for k in range(768)]
for nm, d, sb, ring in (("u1 768 fma acc, 8 chains at v0, sources v64", 0, 64, 8), ("u2 768 fma acc, 8 chains at v128, sources v192", 128, 192, 8),
                        ("u3 768 fma acc, 8 chains at v232, sources v248", 232, 248, 8), ("u11 768 fma acc, 64 chains at v0, sources v192", 0, 192, 64),
                        ("u12 768 fma acc, 120 chains at v0, sources v248", 0, 248, 120)):
    K[nm] = acc_at(d, sb, ring); BIG.add(nm)
for nm0 in ("b0 384 fma baseline", "b15 768 fma VOP3 (8-byte)", "p2 planned PG step x48 (1200 instr, 9.6 KB of code)", "b3 384 fma + 96 s_nop 0",
            "b5 384 fma + 24 ds_read_b128 (+wait at end)", "b6 384 fma + 24 ds_write_b128", "b1 384 fma + 96 v_mov_b64"):
    K["x" + nm0] = K[nm0]; OPTS["x" + nm0] = (126, 0)
for nvv, naa in ((128, 16), (252, 16), (252, 0), (120, 0)):
    nm = f"w 768 fma acc 8 chains at v0, kernel clobbers {nvv} VGPRs + {naa} AGPRs"
    K[nm] = acc_at(0, 64, 8); OPTS[nm] = (nvv, naa)
nm = "p4 planned PG step, 20 steps on distinct registers x2 (1000 instr)"
K[nm] = [ins for r in range(2) for i in range(20) for ins in pg_step_wide(i)]; BIG.add(nm)
ITER = 2000
def gen():
    o = []
    o.append("#include <hip/hip_runtime.h>\n#include <cstdio>\n#include <cstring>\n#ifndef LDS_DOUBLES\n#define LDS_DOUBLES 2048   // 16 KB: four workgroups (one wavefront each) per CU; -DLDS_DOUBLES=8192: two\n#endif\n")
    names = list(K)
    for idx, name in enumerate(names):
        body = K[name]
        nv_, na_ = OPTS.get(name, (252, 64) if name in BIG else (128, 16))
        o.append(f"// {name}\n__global__ __launch_bounds__({512 if na_ == 0 else 64}) void k{idx}(double* out, unsigned long long* stamps, int iters) {{\n  unsigned long long c0, r0, c1, r1;\n  __shared__ double lds[LDS_DOUBLES];\n  lds[threadIdx.x] = 0.25; lds[threadIdx.x + LDS_DOUBLES / 2] = 0.5;\n  double r = 0;\n  asm volatile(\n")
        nv, nacc = OPTS.get(name, (252, 64) if name in BIG else (128, 16))
        pro = []
        for j in range(0, nv, 2):
            pro.append(f"v_cvt_f64_i32 v[{j}:{j+1}], %6")            # lane id as a double
            pro.append(f"v_fma_f64 v[{j}:{j+1}], v[{j}:{j+1}], 0.5, 0.5")
            pro.append(f"v_ldexp_f64 v[{j}:{j+1}], v[{j}:{j+1}], -8")   # (lane/2 + 0.5) / 256: in (0, 0.13)
        pro += ["v_lshlrev_b32 v124, 4, %6", "v_lshlrev_b32 v125, 3, %6", "v_mov_b32 v72, 1", "v_mov_b32 v79, 1",
                "s_mov_b32 s20, 0", "s_mov_b32 s21, 0x3fe00000", "s_mov_b32 s22, 0", "s_mov_b32 s23, 0x3fd00000",
                "s_mov_b32 s24, 0", "s_mov_b32 s25, 0x3fb00000", "s_mov_b32 s26, 0", "s_mov_b32 s27, 0x3fc00000", "s_mov_b32 s29, 7",
                "s_mov_b32 s28, %5", "s_mov_b64 s[12:13], 0", "s_mov_b64 s[14:15], exec", "s_mov_b32 s16, 0x7fffffff", "s_nop 4"]
        pro += [f"v_accvgpr_write_b32 a{j}, v{j}" for j in range(nacc)]
        lines = pro + ["s_memtime %1", "s_memrealtime %2", "s_waitcnt lgkmcnt(0)"] + ([".p2align 3"] if "aligned" in name else []) + (["s_nop 0"] if "mis-aligned" in name else []) + ["1:"] + body + ["s_sub_u32 s28, s28, 1", "s_cmp_lg_u32 s28, 0", "s_cbranch_scc1 1b", "2:", "s_memtime %3", "s_memrealtime %4", "s_waitcnt lgkmcnt(0)", "s_nop 4", "v_mov_b64 %0, v[0:1]"]
        for l in lines:
            o.append(f'    "{l}\\n"\n')
        clob = ", ".join([f'"v{j}"' for j in range(nv)] + [f'"a{j}"' for j in range(nacc)] + [f'"s{j}"' for j in list(range(12, 17)) + list(range(20, 32))] + ['"vcc"', '"scc"', '"memory"'])
        o.append(f'    : "=v"(r), "=&s"(c0), "=&s"(r0), "=&s"(c1), "=&s"(r1) : "s"(iters), "v"((int)threadIdx.x) : {clob});\n  out[blockIdx.x * 64 + threadIdx.x] = r + lds[threadIdx.x];\n  if (threadIdx.x == 0) {{ stamps[2 * blockIdx.x] = c1 - c0; stamps[2 * blockIdx.x + 1] = r1 - r0; }}\n}}\n')
    o.append("typedef void (*kern_t)(double*, unsigned long long*, int);\nstruct Ent { const char* name; kern_t k; int n; int maxthreads; };\nstatic Ent ents[] = {\n")
    for idx, name in enumerate(names):
        nv_, na_ = OPTS.get(name, (252, 64) if name in BIG else (128, 16))
        o.append(f'  {{"{name}", k{idx}, {len(K[name])}, {512 if na_ == 0 else 64}}},\n')
    o.append("};\nint main(int argc, char** argv) {\n  double* d; (void)hipMalloc(&d, 4096 * 64 * 8);\n  unsigned long long* st; (void)hipMalloc(&st, 4096 * 16);\n  static unsigned long long hs[8192];\n"
             f"  const int iters = {ITER};\n"
             "  for (auto& e : ents) {\n    if (argc > 1) { bool hit = false; for (int a = 1; a < argc; ++a) hit = hit || strstr(e.name, argv[a]) == e.name; if (!hit) continue; }\n    printf(\"%-72s\", e.name);\n    for (int cfg = 0; cfg < 3; ++cfg) {\n      const int blocks = cfg == 0 ? 1024 : (cfg == 1 ? 2048 : 256), threads = cfg == 2 ? 512 : 64;   // cfg 2: one workgroup of 8 wavefronts per CU = exactly 2 per SIMD\n      const double norm = (double)blocks * threads / 65536.0;\n      if (threads > e.maxthreads) continue;\n"
             "      hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);\n"
             "      e.k<<<blocks, threads>>>(d, st, 10); (void)hipDeviceSynchronize();\n      (void)hipEventRecord(e0);\n      e.k<<<blocks, threads>>>(d, st, iters);\n"
             "      (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);\n      float ms; (void)hipEventElapsedTime(&ms, e0, e1);\n"
             "      (void)hipMemcpy(hs, st, blocks * 16, hipMemcpyDeviceToHost);\n      double cyc = 0, ghz = 0; for (int b = 0; b < blocks; ++b) { cyc += (double)hs[2 * b]; ghz += (double)hs[2 * b] / (double)hs[2 * b + 1] * 0.1; }\n"
             "      printf(\"  %8.1f ns/body %6.3f ns/instr %5.2f cyc/instr @ %.2f GHz\", ms * 1e6 / iters / norm, ms * 1e6 / iters / e.n / norm, cyc / blocks / iters / e.n, ghz / blocks);\n    }\n"
             "    printf(\"  (%d instr; 1024 x 64 | 2048 x 64 | 256 x 512 threads)\\n\", e.n);\n  }\n  return 0;\n}\n")
    return "".join(o)

if __name__ == "__main__":
    sys.stdout.write(gen())
