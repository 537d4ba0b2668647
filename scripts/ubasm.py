#!/usr/bin/env python3
"""Instruction streams of the fp64 LANE_FMA projected-gradient iteration (csrc/mpc_ub_model.h, operation for operation)
with hand-assigned registers -- shared by scripts/gen_ub_pg_asm.py (the shipped kernel's loop) and
scripts/probes/issue_forms.py (the same stream timed in isolation).

Register plan (one wavefront per SIMD: 256 VGPRs + 256 AGPRs):
  * x (2H values) in place;
  * per register-resident step i two pairs of doubles, V[i] and W[i], that swap roles every iteration: the pair holding
    (Z[i], Y[i]) of the forward pass is dead once the backward sweep has consumed it and receives the new momentum
    vector v[i]; the pair holding the old v[i] is dead once the momentum difference is formed and takes (Z[i], Y[i]) in
    the next forward pass.  Two iterations (halves A and B) are written out, so no value is ever copied.
  * v of the last NA steps lives in AGPRs (2 + 2 transfers per value and iteration), their W pairs never swap.  NA = 3 is the
    least that fits: 3 x 8 + 17 x 12 registers of state, 18 of constants, 10 temporaries = 256.
"""


class Reg:
    """a 64-bit VGPR value: a hard-coded pair (sub-registers nameable) or a named asm operand"""
    def __init__(self, s, lo=None, hi=None):
        self.s, self.lo, self.hi = s, lo, hi
    def __str__(self): return self.s
    def neg(self): return "-" + self.s
    def abs(self): return "|" + self.s + "|"


def hard(r):
    assert r % 2 == 0
    return Reg(f"v[{r}:{r + 1}]", f"v{r}", f"v{r + 1}")


def op(name): return Reg(f"%[{name}]")


class Plan:
    def __init__(self, H=20, NA=4, batch=3, all_hard=False, order="plain"):
        self.H, self.NA, self.NREG, self.BATCH = H, NA, H - NA, batch
        self.order = order
        H_, NREG = H, H - NA
        nclob = 4 * H + 10
        self.BASE = 256 - nclob if not all_hard else None
        if all_hard:
            # x | V | consts | W | T0 T1 n0 n1 acc
            r = 0
            self.X = [hard(r + 2 * q) for q in range(2 * H)]; r += 4 * H
            self.V = [hard(r + 2 * q) for q in range(2 * NREG)]; r += 4 * NREG
            names = ("ca", "cc", "cas", "ccs", "cq", "cil", "cb", "cgl0", "cgl1")
            self.C = {n: hard(r + 2 * j) for j, n in enumerate(names)}; r += 2 * len(names)
            base = r
        else:
            self.X = [op(f"x{q}") for q in range(2 * H)]
            self.V = [op(f"v{q}") for q in range(2 * NREG)]
            self.C = {n: op(n) for n in ("ca", "cc", "cas", "ccs", "cq", "cil", "cb", "cgl0", "cgl1")}
            base = self.BASE
        self.Wz = [hard(base + 4 * i) for i in range(H)]
        self.Wy = [hard(base + 4 * i + 2) for i in range(H)]
        t = base + 4 * H
        self.t0, self.t1, self.n0, self.n1, self.acc = (hard(t + 2 * j) for j in range(5))
        self.top = t + 10
        assert self.top <= 256, self.top
        self.first_clobber = base
        # scalar / AGPR operand names (overridable for the all-hard microbenchmark)
        self.S = {n: f"%[{n}]" for n in ("sgq0", "sgq1", "sgrs0", "sgrs1", "slo1", "sgeps", "shave", "sexec", "sleft")}
        self.A = {}
        for q in range(2 * NREG, 2 * H):
            self.A[f"av{q}lo"] = f"%[av{q}lo]"; self.A[f"av{q}hi"] = f"%[av{q}hi]"
        self.A["z0lo"], self.A["z0hi"] = "%[z0lo]", "%[z0hi]"

    # ---- mpc_ub_model.h Unit::fwd_init / fwd (mpc.h:275-277): (Z, Y) of step i into the step's free pair
    def zy(self, half, i):
        if i >= self.NREG or half == "A": return self.Wz[i], self.Wy[i]
        return self.V[2 * i], self.V[2 * i + 1]

    def forward(self, half):
        C, S, A, t0 = self.C, self.S, self.A, self.t0
        o = []
        for i in range(self.H):
            z, y = self.zy(half, i)
            x0, x1 = self.X[2 * i], self.X[2 * i + 1]
            if i == 0:
                o += [f"v_accvgpr_read_b32 {t0.lo}, {A['z0lo']}", f"v_accvgpr_read_b32 {t0.hi}, {A['z0hi']}",   # Z[-1] = z0
                      f"v_fma_f64 {z}, {C['ca']}, {S['slo1']}, {t0}",           # fma(a, Y[-1] = lo1, Z[-1])
                      f"v_fma_f64 {y}, {C['ccs'].neg()}, {x1}, {S['slo1']}",
                      f"v_fma_f64 {z}, {C['cas']}, {x1}, {z}",
                      f"v_fma_f64 {y}, {C['ccs']}, {x0}, {y}"]
            else:
                zp, yp = self.zy(half, i - 1)
                o += [f"v_fma_f64 {z}, {C['ca']}, {yp}, {zp}",
                      f"v_fma_f64 {y}, {C['ccs'].neg()}, {x1}, {yp}",
                      f"v_fma_f64 {z}, {C['cas']}, {x1}, {z}",
                      f"v_fma_f64 {y}, {C['ccs']}, {x0}, {y}"]
        return o

    # ---- the backward sweep: Unit::bwd_last / bwd (mpc.h:278-281), gradient (mpc.h:283), projected step (mpc.h:342),
    #      stop-test term, momentum step (mpc.h:343); at step 0 the stop test and the end of the iteration
    #      `checks` (steps visited -> label prefix): after that many steps the sweep asks whether the stop test is already
    #      DECIDED for every lane with work -- the running maximum has reached g eps, and mpc.h:310 needs every term below
    #      it -- and if so goes on in the sweep WITHOUT the three stop-test instructions per variable (`notest_from`:
    #      that version, from the given step down, entry labels prefix + step).  Same decisions, fewer instructions.
    def backward(self, half, tail, checks=None, notest_from=None, label=None):
        C, S, A = self.C, self.S, self.A
        t0, t1, n0, n1, acc = self.t0, self.t1, self.n0, self.n1, self.acc
        o = []
        first = True
        test = notest_from is None
        for i in range(self.H - 1 if test else notest_from, -1, -1):
            if not test:
                o += [f"{label}{i}%=:"]
            x0, x1 = self.X[2 * i], self.X[2 * i + 1]
            agpr = i >= self.NREG
            if agpr:
                pz, py, o0, o1 = self.Wz[i], self.Wy[i], t0, t1
                o += [f"v_accvgpr_read_b32 {t0.lo}, {A[f'av{2 * i}lo']}", f"v_accvgpr_read_b32 {t0.hi}, {A[f'av{2 * i}hi']}",
                      f"v_accvgpr_read_b32 {t1.lo}, {A[f'av{2 * i + 1}lo']}", f"v_accvgpr_read_b32 {t1.hi}, {A[f'av{2 * i + 1}hi']}"]
            elif half == "A":
                pz, py, o0, o1 = self.Wz[i], self.Wy[i], self.V[2 * i], self.V[2 * i + 1]
            else:
                pz, py, o0, o1 = self.V[2 * i], self.V[2 * i + 1], self.Wz[i], self.Wy[i]
            last = i == 0
            s0, s1 = (t0, t1) if last else (x0, x1)     # step 0 keeps x[0], x[1] for the lanes that stop
            d1 = t0 if last else o0                     # where df1 is formed
            if i == self.H - 1:
                bw = [f"v_mul_f64 {n0}, {S['sgq0']}, {pz}",
                      f"v_fma_f64 {n1}, {S['sgq1']}, {py}, {C['cq'].neg()}"]
            else:
                bw = [f"v_fma_f64 {py}, {S['sgq1']}, {py}, {C['cq'].neg()}",     # e1
                      f"v_fma_f64 {n1}, {C['ca']}, {n0}, {n1}",
                      f"v_add_f64 {n1}, {n1}, {py}",                            # t1 = fma(a, n0, n1) + e1
                      f"v_fma_f64 {n0}, {S['sgq0']}, {pz}, {n0}"]
            o += bw
            o += [f"v_fma_f64 {py}, {S['sgrs0']}, {x0}, {C['cgl0']}",
                  f"v_fma_f64 {py}, {C['cc']}, {n1}, {py}",                 # df0
                  f"v_fma_f64 {pz}, {C['cil'].neg()}, {py}, {x0} clamp"]    # vn0
            if test:
                o += [f"v_add_f64 {s0}, {x0}, {pz.neg()}"]
                if first:
                    o += [f"v_min_f64 {acc}, {py.abs()}, {s0.abs()}"]
                    first = False
                else:
                    o += [f"v_min_f64 {s0}, {py.abs()}, {s0.abs()}", f"v_max_f64 {acc}, {acc}, {s0}"]
            if not last:
                o += [f"v_add_f64 {o0}, {pz}, {o0.neg()}",             # vn0 - vold0
                      f"v_fma_f64 {x0}, {C['cb']}, {o0}, {pz} clamp"]
                if agpr:
                    o += [f"v_accvgpr_write_b32 {A[f'av{2 * i}lo']}, {pz.lo}", f"v_accvgpr_write_b32 {A[f'av{2 * i}hi']}, {pz.hi}"]
            o += [f"v_fma_f64 {d1}, {S['sgrs1']}, {x1}, {C['cgl1']}",
                  f"v_fma_f64 {d1}, {C['cc'].neg()}, {n1}, {d1}",
                  f"v_fma_f64 {d1}, {C['ca']}, {n0}, {d1}",                 # df1
                  f"v_fma_f64 {py}, {C['cil'].neg()}, {d1}, {x1} clamp"]    # vn1
            if test:
                o += [f"v_add_f64 {s1}, {x1}, {py.neg()}",
                      f"v_min_f64 {s1}, {d1.abs()}, {s1.abs()}",
                      f"v_max_f64 {acc}, {acc}, {s1}"]
            if not last:
                o += [f"v_add_f64 {o1}, {py}, {o1.neg()}",
                      f"v_fma_f64 {x1}, {C['cb']}, {o1}, {py} clamp"]
                if agpr:
                    o += [f"v_accvgpr_write_b32 {A[f'av{2 * i + 1}lo']}, {py.lo}", f"v_accvgpr_write_b32 {A[f'av{2 * i + 1}hi']}, {py.hi}"]
                if test and checks and (self.H - i) in checks:
                    # decided = acc >= g eps (false for a NaN: such a lane stays undecided); every lane with work decided -> no stop this iteration
                    o += [f"v_cmp_le_f64_e64 vcc, {S['sgeps']}, {acc}",
                          f"s_andn2_b64 vcc, {S['shave']}, vcc",
                          f"s_cbranch_scc0 {checks[self.H - i]}{i - 1}%="]
            elif not test:
                o += [f"v_add_f64 {o0}, {pz}, {o0.neg()}",
                      f"v_fma_f64 {x0}, {C['cb']}, {o0}, {pz} clamp",
                      f"v_add_f64 {o1}, {py}, {o1.neg()}",
                      f"v_fma_f64 {x1}, {C['cb']}, {o1}, {py} clamp",
                      f"s_sub_u32 {S['sleft']}, {S['sleft']}, 1"]
                o += tail
            else:
                # (e64: the loop's 4-byte instructions must come in PAIRS -- an 8-byte instruction that starts on an odd dword costs a
                #  lone wavefront 5 cycles instead of 4, scripts/probes/issue_forms.py)
                o += [f"v_cmp_gt_f64_e64 vcc, {S['sgeps']}, {acc}",         # max_df < g eps (mpc.h:310)
                      f"s_and_b64 vcc, vcc, {S['shave']}",
                      f"s_andn2_b64 exec, {S['sexec']}, vcc",
                      f"v_add_f64 {o0}, {pz}, {o0.neg()}",
                      f"v_fma_f64 {x0}, {C['cb']}, {o0}, {pz} clamp",
                      f"v_add_f64 {o1}, {py}, {o1.neg()}",
                      f"v_fma_f64 {x1}, {C['cb']}, {o1}, {py} clamp",
                      f"s_mov_b64 exec, {S['sexec']}",
                      f"s_sub_u32 {S['sleft']}, {S['sleft']}, 1"]           # borrow (SCC): the smallest budget of the wavefront is spent (mpc.h:271)
                o += tail
        return o

    def iteration(self, half, tail):
        return self.forward(half) + self.backward(half, tail)
