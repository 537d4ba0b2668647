#!/usr/bin/env python3
"""Generates trajectory_controller_amd/csrc/mpc_ub_pg_asm.h: the persistent projected-gradient kernel of the fp64
LANE_FMA family at N = 20 (both inputs sharing their bounds, screened stop test: ub_pg_kernel<double, 20, true, 2> of
mpc_ub.h) as ONE inline-asm statement -- iteration loop, stop events, refill passes and result stores -- with every
vector register assigned by hand.

Why by hand: a lone wavefront per SIMD pays ~4 cycles for EVERY instruction it issues, whatever it is (fp64
arithmetic, register moves, AGPR transfers, scalar instructions: scripts/probes/issue_forms.py,
profiles/r05_issue_forms*.txt), 5 when an 8-byte instruction starts on an odd dword, and the compiler's version of
this loop spends 193 of its 708 instructions on register shuffles and control flow and sits on arbitrary addresses.

The arithmetic is that of csrc/mpc_ub_model.h operation for operation (the CPU model the GPU tests hold the kernel to
bit for bit); where values live is decided in scripts/ubasm.py (the iteration) and here (everything around it):

  * the stop test (mpc.h:310-311) is evaluated BEFORE step 0's momentum update, which runs under an EXEC mask that
    leaves out the lanes that stop: their x[0], x[1] are dlib's answer (the controls before the update), stored at once
    by the stop block; the loop goes on until a refill pass is due (BATCH lanes wait) or no lane has work;
  * the iteration count is wave-uniform (SGPR count-down to the earliest cap of the wavefront's lanes, mpc.h:271); a
    lane's own count is that plus a per-lane base fixed at refill;
  * a refill pass takes its model scalars and step constants from the instance's record, where the coordinate-descent
    kernel left them (same operations, same bits as recomputing them: mpc_ub.h), so it contains no division.

    python scripts/gen_ub_pg_asm.py [NA] > trajectory_controller_amd/csrc/mpc_ub_pg_asm.h
"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ubasm

H = int(sys.argv[6]) if len(sys.argv) > 6 else 20   # 20: the headline kernel (mpc_ub_pg_asm.h); 10: mpc_ub_pg_asm_h10.h (the state fits the VGPRs: NA = 0)
SUFFIX = "" if H == 20 else f"_h{H}"
NA = int(sys.argv[1]) if len(sys.argv) > 1 else 3   # steps H-NA .. H-1 keep v in AGPRs
DIAG = int(sys.argv[2]) if len(sys.argv) > 2 else 0   # diagnostic builds only (scripts/build_asm_variant.sh): the wave-iteration statistic
                                                      # becomes the shader cycles spent in: 1 ticket wait, 2 queue-entry wait, 3 record wait, 4 a whole refill pass, 5 the iteration loop
BATCH = int(sys.argv[7]) if len(sys.argv) > 7 else (3 if H == 20 else 10)   # lanes that must wait before a refill pass is due (N = 10, 1 048 576 instances: 3.94 / 2.54 / 2.19 / 2.24 ms at 3 / 6 / 10 / 16)
RL_BYTES = (2 * H + 1 + 1 + 7 + 1) // 2 * 2 * 8     # LaneRec<double, H>::kLen * 8 (400 at N = 20)
P = ubasm.Plan(H=H, NA=NA, batch=BATCH, all_hard=True)
NREG = P.NREG
X, V, Wz, Wy, t0, t1, n0, n1, acc, C = P.X, P.V, P.Wz, P.Wy, P.t0, P.t1, P.n0, P.n1, P.acc, P.C
# per-lane bookkeeping touched only at stop events and refill passes: instance index, iteration-count base, cap (in wave
# iterations).  With room (NA >= 4) in the last VGPRs; with NA = 3 the loop's state fills all 256 and they live in AGPRs
PARK = P.top > 252
assert P.top <= 256
# (N = 20: the last four of the 256; a horizon whose state leaves room keeps them right above it, so that the statement claims
#  only the registers it uses and several wavefronts fit a SIMD -- at N = 10 the second one hides the refill passes)
_BK = 252 if H == 20 else P.top
NVGPR = 256 if H == 20 else P.top + 4
VK, VBASE, VCAP, VTMP = (f"v{_BK}", f"v{_BK + 1}", f"v{_BK + 2}", f"v{_BK + 3}") if not PARK else (None, None, None, n0.lo)
# AGPRs: v of the AGPR steps, z0
AG = {}
_n = 3 if PARK else 0
AK, ABASE, ACAP = "a0", "a1", "a2"
for q in range(2 * NREG, 2 * H):
    AG[f"av{q}lo"], AG[f"av{q}hi"] = f"a{_n}", f"a{_n + 1}"; _n += 2
AG["z0lo"], AG["z0hi"] = f"a{_n}", f"a{_n + 1}"; _n += 2
# CHUNK > 0: tickets are drawn CHUNK at a time (one device-scope atomic and one load of queue entries per CHUNK / BATCH refill
# passes instead of one each per pass: ~1 770 and ~490 cycles of waiting, scripts/probes/asm_diag.sh); the chunk's queue entries
# wait in an AGPR, lane j = entry j.  Measured SLOWER on one box (profiles/r05_ab_chunk.txt: 4.75 ms with a draw per pass, 4.90 /
# 4.94 / 4.92 with chunks of 6 / 12 / 24 -- a pass that finds fewer tickets than waiting lanes leaves lanes idle, +1.2 % wave
# iterations, and the waiting it saves was not the limit: the kernel is power-bound).  0 (shipped): a ticket draw per pass
CHUNK = int(sys.argv[4]) if len(sys.argv) > 4 else 0
# CHECKS: after that many steps of the backward sweep, ask whether the stop test is already decided for the whole wavefront
# (ubasm.Plan.backward) and if so finish the sweep without its 3 stop-test instructions per variable; "0" = never.  On the
# headline batch the test is decided for all 64 lanes after 2 steps in 48 % of a wavefront's iterations, after 5 in 65 %
# (tests/extended/stop_depth_stats.cpp + stop_depth_sim.py); one box, PG kernel: 4.78-4.80 ms without, 4.50-4.62 with any of
# "1,2,4" "2,4" "2" "3" "2,5" "3,6", "1,2,3,5" and "1,2,3,4,6,9" 2 % behind those (profiles/r05_ab_checks.txt)
CHECKS = [int(c) for c in (sys.argv[5] if len(sys.argv) > 5 else "2,5").split(",") if int(c) > 0]
AORD = f"a{_n}"
if CHUNK: _n += 1
N_AGPR = _n
P.A = AG
P.S = {n: f"%[{n}]" for n in ("sgq0", "sgq1", "sgrs0", "sgrs1", "slo1", "sgeps", "shave", "sleft")}
EXECMASK = int(sys.argv[3]) if len(sys.argv) > 3 else 0   # 1: the loop runs under EXEC = the lanes that carry an instance (measured SLOWER: 5.14-5.24 ms against 4.82-4.87,
                                                          # profiles/r05_ab_execmask.txt); 0: every lane iterates, idle ones on stale state
P.S["sexec"] = "%[shave]" if EXECMASK else "-1"
W0 = int(Wz[0].lo[1:])


def scratch(j):
    """a pair of the W region (free outside the iteration loop)"""
    return ubasm.hard(W0 + 2 * j)


def publish(which, it_expr):
    """stores the controls of the lanes in EXEC: front[k] = control(0, x[0]), rear[k] = control(1, x[1]) (Unit::control:
    the bounds and the untouched start point come out exactly), iters[k] = it_expr + the lane's base; `which`: 'x' = from
    x[0], x[1]; 'zero' = 0, 0 (non-finite inputs).  Uses t0, t1, acc, VTMP, vcc, stmp."""
    kreg, breg = (VK, VBASE) if not PARK else (n0.hi, n1.lo)
    o = ([f"v_accvgpr_read_b32 {kreg}, {AK}", f"v_accvgpr_read_b32 {breg}, {ABASE}"] if PARK else []) + [f"v_lshlrev_b32 {VTMP}, 3, {kreg}"]
    if which == "zero":
        o += [f"v_mov_b64 {t0}, 0", f"v_mov_b64 {t1}, 0"]
    else:
        o += ["s_mov_b64 %[stmp], exec"]
        for j, (xx, tt) in enumerate(((X[0], t0), (X[1], t1))):
            o += [f"v_mov_b64 {tt}, %[slo{j}]",
                  f"v_fma_f64 {tt}, %[ss{j}], {xx}, {tt}",                 # fma(s, x, lo)
                  f"v_cmp_eq_f64_e64 vcc, {xx}, %[sxz{j}]", "s_and_b64 exec, %[stmp], vcc", f"v_mov_b64 {tt}, 0",
                  f"v_cmp_eq_f64_e64 vcc, {xx}, 1.0", "s_and_b64 exec, %[stmp], vcc", f"v_mov_b64 {tt}, %[shi{j}]",
                  "s_mov_b64 exec, %[stmp]"]
    o += [f"global_store_dwordx2 {VTMP}, {t0}, %[pfront]", f"global_store_dwordx2 {VTMP}, {t1}, %[prear]",
          "s_cmp_eq_u64 %[piters], 0", "s_cbranch_scc1 1f",
          f"v_lshlrev_b32 {VTMP}, 2, {kreg}", f"v_add_u32 {acc.lo}, {it_expr}, {breg}",
          f"global_store_dword {VTMP}, {acc.lo}, %[piters]", "1:"]
    return o


def stop_block(cont_label, exit_label):
    """lanes whose stop test fired (vcc): store their answer -- x[0], x[1] before this iteration's update; iteration
    count = base + wave iterations - 1 --, take them out of `have`; go on unless a refill pass is due, no lane has work
    left, or the cap fell in the same iteration"""
    o = ["s_cselect_b32 %[scapf], 1, 0",                      # cap pending (SCC of the count-down)
         "s_mov_b64 %[sstop], vcc",
         "s_mov_b64 exec, vcc",
         "s_sub_u32 %[sa], %[scap], %[sleft]", "s_sub_u32 %[sa], %[sa], 2"]   # wave iterations run (cap - 1 - left), less one
    o += publish("x", "%[sa]")
    o += ["s_andn2_b64 %[shave], %[shave], %[sstop]",
          f"s_mov_b64 exec, {P.S['sexec']}",
          "s_or_b64 %[stmp], %[shave], %[sexh]", "s_not_b64 %[stmp], %[stmp]",     # lanes waiting for a refill
          "s_bcnt1_i32_b64 %[sb], %[stmp]",
          "s_cmp_lg_u32 %[scapf], 0", f"s_cbranch_scc1 {exit_label}",
          f"s_cmp_ge_u32 %[sb], {BATCH}", f"s_cbranch_scc1 {exit_label}",
          "s_cmp_eq_u64 %[shave], 0", f"s_cbranch_scc1 {exit_label}",
          f"s_branch {cont_label}"]
    return o


def stamp_begin(k):
    return ["s_memtime s[96:97]", "s_waitcnt lgkmcnt(0)"] if DIAG == k else []


def stamp_end(k):
    return (["s_memtime s[98:99]", "s_waitcnt lgkmcnt(0)", "s_sub_u32 s98, s98, s96", "s_add_u32 %[sdiag], %[sdiag], s98"]
            if DIAG == k else [])


def gen_body():
    sc = [scratch(j) for j in range(14)]   # refill scratch pairs
    vk, vbase, vcap = (VK, VBASE, VCAP) if not PARK else (sc[12].lo, sc[12].hi, sc[13].lo)   # (refill's working copies)
    o = ["s_mov_b32 %[sdiag], 0"] if DIAG else []
    if DIAG == 6: o += ["s_memtime s[96:97]", "s_waitcnt lgkmcnt(0)"]
    if DIAG == 7: o += ["s_memrealtime s[96:97]", "s_waitcnt lgkmcnt(0)"]
    # ---------------- prologue: state of every lane to zero (idle lanes iterate on it harmlessly)
    o += ["s_mov_b64 %[shave], 0", "s_mov_b64 %[sexh], 0", "s_mov_b32 %[sit], 0", "s_mov_b32 %[srefills], 0",
          "s_mov_b32 %[sflags], 0", "s_mov_b32 %[scap], -1"]
    if CHUNK:
        o += ["s_mov_b32 %[scnext], 0", "s_mov_b32 %[scend], 0", "s_mov_b32 %[scbase], 0", f"v_accvgpr_write_b32 {AORD}, 0"]
    for r in X + V + list(C.values()):
        o += [f"v_mov_b64 {r}, 0"]
    for a in AG.values():
        o += [f"v_accvgpr_write_b32 {a}, 0"]
    if PARK:
        o += [f"v_accvgpr_write_b32 {AK}, 0", f"v_accvgpr_write_b32 {ABASE}, 0", f"v_accvgpr_write_b32 {ACAP}, -1"]
    else:
        o += [f"v_mov_b32 {VK}, 0", f"v_mov_b32 {VBASE}, 0", f"v_mov_b32 {VCAP}, -1"]
    o += [f"v_mov_b64 {C['cgl0']}, %[sgrl0]", f"v_mov_b64 {C['cgl1']}, %[sgrl1]"]
    # ---------------- main: refill when due, else iterate
    o += ["MAIN%=:",
          "s_or_b64 %[stmp], %[shave], %[sexh]", "s_not_b64 %[swant], %[stmp]",
          "s_cmp_eq_u64 %[swant], 0", "s_cbranch_scc1 NOWANT%=",
          "s_bcnt1_i32_b64 %[scnt], %[swant]",
          f"s_cmp_ge_u32 %[scnt], {BATCH}", "s_cbranch_scc1 REFILL%=",
          "s_cmp_eq_u64 %[shave], 0", "s_cbranch_scc1 REFILL%=",
          "s_branch ITER%=",
          "NOWANT%=:",
          "s_cmp_eq_u64 %[shave], 0", "s_cbranch_scc1 DONE%=",
          "s_branch ITER%="]
    # ---------------- refill pass (lane_pg_fused_kernel's protocol: one atomic for the wavefront's tickets)
    if not CHUNK:
        o += ["REFILL%=:"] + stamp_begin(4) + [
              "s_add_u32 %[srefills], %[srefills], 1",
              "s_ff1_i32_b64 %[sa], %[swant]", "s_lshl_b64 exec, 1, %[sa]",
              f"v_mov_b32 {sc[0].lo}, %[scnt]", f"v_mov_b32 {sc[0].hi}, 0"] + stamp_begin(1) + [
              f"global_atomic_add {sc[1].lo}, {sc[0].hi}, {sc[0].lo}, %[pticket] sc0",
              "s_waitcnt vmcnt(0)"] + stamp_end(1) + ["s_nop 1",
              f"v_readfirstlane_b32 %[sfirst], {sc[1].lo}",
              "s_mov_b64 exec, %[swant]",
              f"v_mbcnt_lo_u32_b32 {sc[0].lo}, exec_lo, 0", f"v_mbcnt_hi_u32_b32 {sc[0].lo}, exec_hi, {sc[0].lo}",   # rank among the waiting lanes
              f"v_add_u32 {sc[0].lo}, %[sfirst], {sc[0].lo}",          # ticket
              f"v_cmp_le_u32_e64 vcc, %[snq], {sc[0].lo}",              # past the queue's end: exhausted
              "s_or_b64 %[sexh], %[sexh], vcc",
              "s_andn2_b64 %[snew], exec, vcc",
              "s_mov_b64 exec, %[snew]",
              "s_cbranch_execz RDONE%=",
              f"v_lshlrev_b32 {sc[0].lo}, 2, {sc[0].lo}"] + stamp_begin(2) + [
              f"global_load_dword {vk}, {sc[0].lo}, %[porder]",
              "s_waitcnt vmcnt(0)"] + stamp_end(2) + ([f"v_accvgpr_write_b32 {AK}, {vk}"] if PARK else [])
    else:
        o += ["REFILL%=:"] + stamp_begin(4) + [
              "s_add_u32 %[srefills], %[srefills], 1",
              "s_sub_u32 %[sb], %[scend], %[scnext]",                    # tickets left in the wavefront's chunk
              "s_cmp_lg_u32 %[sb], 0", "s_cbranch_scc1 RCHUNK%=",
              # ---- a new chunk: max(CHUNK, waiting lanes) tickets with one atomic (lane_pg_fused_kernel's protocol)
              f"s_max_u32 %[sb], %[scnt], {CHUNK}",
              "s_ff1_i32_b64 %[sa], %[swant]", "s_lshl_b64 exec, 1, %[sa]",
              f"v_mov_b32 {sc[0].lo}, %[sb]", f"v_mov_b32 {sc[0].hi}, 0"] + stamp_begin(1) + [
              f"global_atomic_add {sc[1].lo}, {sc[0].hi}, {sc[0].lo}, %[pticket] sc0",
              "s_waitcnt vmcnt(0)"] + stamp_end(1) + ["s_nop 1",
              f"v_readfirstlane_b32 %[scbase], {sc[1].lo}",
              "s_mov_b32 %[scnext], %[scbase]",
              "s_add_u32 %[scend], %[scbase], %[sb]",
              "s_min_u32 %[scend], %[scend], %[snq]", "s_max_u32 %[scend], %[scend], %[scnext]",   # the part of it inside the queue
              "s_sub_u32 %[sb], %[scend], %[scnext]",
              "s_cmp_eq_u32 %[sb], 0", "s_cbranch_scc1 REXH%=",
              # its queue entries: lane j < sb loads entry j
              "s_mov_b64 exec, -1",
              f"v_mbcnt_lo_u32_b32 {sc[0].lo}, -1, 0", f"v_mbcnt_hi_u32_b32 {sc[0].lo}, -1, {sc[0].lo}",   # lane number
              f"v_cmp_gt_u32_e64 vcc, %[sb], {sc[0].lo}", "s_mov_b64 exec, vcc",
              f"v_add_u32 {sc[0].lo}, %[scbase], {sc[0].lo}", f"v_lshlrev_b32 {sc[0].lo}, 2, {sc[0].lo}"] + stamp_begin(2) + [
              f"global_load_dword {sc[1].lo}, {sc[0].lo}, %[porder]",
              "s_waitcnt vmcnt(0)"] + stamp_end(2) + [
              f"v_accvgpr_write_b32 {AORD}, {sc[1].lo}",
              "s_branch RCHUNK%=",
              "REXH%=:",                                                  # the queue is empty: every waiting lane is exhausted
              "s_or_b64 %[sexh], %[sexh], %[swant]", "s_branch RDONE%=",
              # ---- min(left in the chunk, waiting lanes) lanes take the chunk's next tickets
              "RCHUNK%=:",
              "s_min_u32 %[sb], %[sb], %[scnt]",
              "s_mov_b64 exec, %[swant]",
              f"v_mbcnt_lo_u32_b32 {sc[0].lo}, exec_lo, 0", f"v_mbcnt_hi_u32_b32 {sc[0].lo}, exec_hi, {sc[0].lo}",   # rank among the waiting lanes
              f"v_cmp_gt_u32_e64 vcc, %[sb], {sc[0].lo}",
              "s_and_b64 %[snew], exec, vcc",
              "s_sub_u32 %[sa], %[scnext], %[scbase]",
              f"v_add_u32 {sc[0].lo}, %[sa], {sc[0].lo}", f"v_lshlrev_b32 {sc[0].lo}, 2, {sc[0].lo}",   # byte address of the lane that holds the entry
              "s_add_u32 %[scnext], %[scnext], %[sb]",
              "s_mov_b64 exec, -1",                                       # (ds_bpermute reads zero from a source lane outside EXEC)
              f"v_accvgpr_read_b32 {sc[1].lo}, {AORD}", "s_nop 1",
              f"ds_bpermute_b32 {sc[1].hi}, {sc[0].lo}, {sc[1].lo}",
              "s_waitcnt lgkmcnt(0)",
              "s_mov_b64 exec, %[snew]",
              f"v_mov_b32 {vk}, {sc[1].hi}"] + ([f"v_accvgpr_write_b32 {AK}, {vk}"] if PARK else [])
    o += [f"v_mov_b32 {sc[0].hi}, {RL_BYTES}",
          f"v_mad_u64_u32 {sc[1]}, vcc, {vk}, {sc[0].hi}, %[precs]"]      # &recs[k * RL]
    o += stamp_begin(3)
    for i in range(H):
        o += [f"global_load_dwordx4 v[{4 * i}:{4 * i + 3}], {sc[1]}, off offset:{16 * i}"]   # x[2i], x[2i+1]
    # lambda | meta ; il0 il1 ; beta a ; c ty ; tphi
    r4 = lambda j: f"v[{W0 + 4 + 4 * j}:{W0 + 7 + 4 * j}]"
    o += [f"global_load_dwordx4 {r4(0)}, {sc[1]}, off offset:{16 * H}",
          f"global_load_dwordx4 {r4(1)}, {sc[1]}, off offset:{16 * H + 16}",
          f"global_load_dwordx4 {r4(2)}, {sc[1]}, off offset:{16 * H + 32}",
          f"global_load_dwordx4 {r4(3)}, {sc[1]}, off offset:{16 * H + 48}",
          f"global_load_dwordx2 {sc[10]}, {sc[1]}, off offset:{16 * H + 64}",
          "s_waitcnt vmcnt(0)"] + stamp_end(3)
    lam, meta, il0, il1_, beta, ra, rc, ty, tphi = sc[2], sc[3], sc[4], sc[5], sc[6], sc[7], sc[8], sc[9], sc[10]
    o += [f"v_mov_b64 {C['cil']}, {il0}", f"v_mov_b64 {C['cb']}, {beta}",           # pg_constants (mpc.h:342-343), from the record
          f"v_mov_b64 {C['ca']}, {ra}", f"v_mov_b64 {C['cc']}, {rc}",               # Unit::set_instance_ac
          f"v_mul_f64 {C['cas']}, {ra}, %[ss1]", f"v_mul_f64 {C['ccs']}, {rc}, %[ss0]",
          f"v_add_f64 {sc[11]}, 0, {ty.neg()}",                                      # z0 = 0 - ty
          f"v_accvgpr_write_b32 {AG['z0lo']}, {sc[11].lo}", f"v_accvgpr_write_b32 {AG['z0hi']}, {sc[11].hi}",
          f"v_add_f64 {sc[11]}, %[slo1], {tphi}", f"v_mul_f64 {C['cq']}, %[sgq1], {sc[11]}",   # q1th = gq1 (lo1 + tphi)
          f"v_subrev_u32 {vbase}, %[sit], {meta.lo}",                                # lane's count = base + wave iterations
          f"v_sub_u32 {vcap}, %[smaxit], {vbase}"]                                   # wave iteration at which the lane reaches max_iter
    if PARK:
        o += [f"v_accvgpr_write_b32 {ABASE}, {vbase}", f"v_accvgpr_write_b32 {ACAP}, {vcap}"]
    # v := x where the coordinate-descent phase ended on its last iteration (mpc.h:330-334), else the start point u = 0
    o += [f"v_mov_b64 {t0}, %[sxz0]", f"v_mov_b64 {t1}, %[sxz1]"]
    for q in range(2 * NREG):
        o += [f"v_mov_b64 {V[q]}, {t0 if q % 2 == 0 else t1}"]
    for q in range(2 * NREG, 2 * H):
        tt = t0 if q % 2 == 0 else t1
        o += [f"v_accvgpr_write_b32 {AG[f'av{q}lo']}, {tt.lo}", f"v_accvgpr_write_b32 {AG[f'av{q}hi']}, {tt.hi}"]
    o += [f"v_and_b32 {sc[11].lo}, 2, {meta.hi}", f"v_cmp_ne_u32_e64 vcc, 0, {sc[11].lo}",
          "s_and_b64 exec, %[snew], vcc", "s_cbranch_execz NOVINIT%="]
    for q in range(2 * NREG):
        o += [f"v_mov_b64 {V[q]}, {X[q]}"]
    for q in range(2 * NREG, 2 * H):
        o += [f"v_accvgpr_write_b32 {AG[f'av{q}lo']}, {X[q].lo}", f"v_accvgpr_write_b32 {AG[f'av{q}hi']}, {X[q].hi}"]
    o += ["NOVINIT%=:", "s_mov_b64 exec, %[snew]"]
    # records that are already complete (the coordinate-descent kernel publishes those itself and keeps them out of the
    # queue; kept for a queue that holds one): stopped, or at the cap
    o += [f"v_and_b32 {sc[11].lo}, 1, {meta.hi}", f"v_cmp_ne_u32_e64 vcc, 0, {sc[11].lo}",     # kMetaStopped
          f"v_cmp_le_u32_e64 %[stmp], %[smaxit], {meta.lo}",                                   # iter >= max_iter
          "s_or_b64 %[sdonenow], vcc, %[stmp]", "s_and_b64 %[sdonenow], %[sdonenow], %[snew]",
          "s_cmp_eq_u64 %[sdonenow], 0", "s_cbranch_scc1 RHAVE%=",
          # (rare path)
          f"v_and_b32 {sc[11].lo}, 4, {meta.hi}", f"v_cmp_ne_u32_e64 %[sstop], 0, {sc[11].lo}",    # kMetaNonFinite
          "s_and_b64 %[sstop], %[sstop], %[sdonenow]",
          "s_cmp_lg_u64 %[sstop], 0", "s_cselect_b32 %[sa], 1, 0", "s_or_b32 %[sflags], %[sflags], %[sa]",
          "s_andn2_b64 %[stmp], %[stmp], vcc", "s_and_b64 %[stmp], %[stmp], %[sdonenow]",            # at the cap without having stopped
          "s_cmp_lg_u64 %[stmp], 0", "s_cselect_b32 %[sa], 2, 0", "s_or_b32 %[sflags], %[sflags], %[sa]",
          "s_mov_b64 exec, %[sstop]", "s_cbranch_execz RNF%="]
    o += publish("zero", "%[sit]")
    o += ["RNF%=:", "s_andn2_b64 exec, %[sdonenow], %[sstop]", "s_cbranch_execz RHAVE%="]
    o += publish("x", "%[sit]")
    o += ["RHAVE%=:",
          "s_andn2_b64 %[snew], %[snew], %[sdonenow]",
          "s_or_b64 %[shave], %[shave], %[snew]",
          # the wavefront's earliest cap: the new lanes' against what it was (a lane that left may leave it early: CAPCHK recomputes)
          "s_mov_b64 %[stmp], %[snew]",
          "RCAP%=:", "s_cmp_eq_u64 %[stmp], 0", "s_cbranch_scc1 RDONE%=",
          "s_ff1_i32_b64 %[sa], %[stmp]", "s_bitset0_b64 %[stmp], %[sa]",
          f"v_readlane_b32 %[sb], {vcap}, %[sa]", "s_min_u32 %[scap], %[scap], %[sb]",
          "s_branch RCAP%=",
          "RDONE%=:", "s_mov_b64 exec, -1"] + stamp_end(4) + ["s_branch MAIN%="]
    # ---------------- iterate
    tailA = ["s_cbranch_vccnz SA%=", "s_cbranch_scc1 XODD%="]
    tailB = ["s_cbranch_vccnz SB%=", "s_cbranch_scc1 XEVEN%=", "s_branch LA%="]
    o += ["ITER%=:",
          "s_sub_u32 %[sleft], %[scap], %[sit]", "s_sub_u32 %[sleft], %[sleft], 1"] + stamp_begin(5) + [   # iterations to the earliest cap, less one
          f"s_mov_b64 exec, {P.S['sexec']}",
          ".p2align 3", "LA%=:"]
    chk = {k: "NTA" for k in CHECKS}
    o += P.forward("A") + P.backward("A", tailA, checks=chk)
    chk = {k: "NTB" for k in CHECKS}
    o += ["LB%=:"] + P.forward("B") + P.backward("B", tailB, checks=chk)
    o += ["SA%=:"] + stop_block("LB%=", "XODD%=")
    o += ["SB%=:"] + stop_block("LA%=", "XEVEN%=")
    if CHECKS:
        # the rest of the sweep without the stop test (entered from the checks: no lane of the wavefront can stop in this iteration)
        top = H - 1 - min(CHECKS)
        o += [".p2align 3"] + P.backward("A", ["s_cbranch_scc1 XODD%=", "s_branch LB%="], notest_from=top, label="NTA")
        o += [".p2align 3"] + P.backward("B", ["s_cbranch_scc1 XEVEN%=", "s_branch LA%="], notest_from=top, label="NTB")
    o += ["XODD%=:"] + [f"v_mov_b64 {V[q]}, {(Wz if q % 2 == 0 else Wy)[q // 2]}" for q in range(2 * NREG)]
    o += ["XEVEN%=:", "s_mov_b64 exec, -1"] + stamp_end(5) + [
          "s_sub_u32 %[sit], %[scap], %[sleft]", "s_sub_u32 %[sit], %[sit], 1",       # wave iterations so far
          "s_cmp_ge_u32 %[sit], %[scap]", "s_cbranch_scc0 MAIN%="]
    # ---------------- a lane may have reached max_iter (mpc.h:271): publish those, recompute the earliest cap exactly
    capreg = VCAP if not PARK else acc.hi
    o += ["CAPCHK%=:"] + ([f"v_accvgpr_read_b32 {capreg}, {ACAP}", "s_nop 1"] if PARK else []) + [
          f"v_cmp_le_u32_e64 vcc, {capreg}, %[sit]", "s_and_b64 %[sstop], vcc, %[shave]",
          "s_cmp_eq_u64 %[sstop], 0", "s_cbranch_scc1 CAPMIN%=",
          "s_or_b32 %[sflags], %[sflags], 2",
          "s_mov_b64 exec, %[sstop]"]
    o += publish("x", "%[sit]")
    o += ["s_mov_b64 exec, -1", "s_andn2_b64 %[shave], %[shave], %[sstop]",
          "CAPMIN%=:", "s_mov_b32 %[scap], -1", "s_mov_b64 %[stmp], %[shave]",
          "CAPL%=:", "s_cmp_eq_u64 %[stmp], 0", "s_cbranch_scc1 MAIN%=",
          "s_ff1_i32_b64 %[sa], %[stmp]", "s_bitset0_b64 %[stmp], %[sa]",
          f"v_readlane_b32 %[sb], {capreg}, %[sa]", "s_min_u32 %[scap], %[scap], %[sb]",
          "s_branch CAPL%=",
          "DONE%=:", "s_waitcnt vmcnt(0)"]
    if DIAG == 6: o += ["s_memtime s[98:99]", "s_waitcnt lgkmcnt(0)", "s_sub_u32 %[sdiag], s98, s96"]
    if DIAG == 7: o += ["s_memrealtime s[98:99]", "s_waitcnt lgkmcnt(0)", "s_sub_u32 %[sdiag], s98, s96"]
    return o


def gen():
    body = gen_body()
    n_iter = len(P.forward("A")) + len(P.backward("A", ["x", "y"]))
    out = []
    out.append("// GENERATED by scripts/gen_ub_pg_asm.py -- do not edit (make -C csrc regen).  The persistent projected-gradient kernel\n"
               "// of ub_pg_kernel<double, %d, true, 2> (mpc_ub.h) as one asm statement with hand-assigned registers; arithmetic:\n"
               "// mpc_ub_model.h, operation for operation.  %d instructions per iteration%s; v of steps\n"
               "// %s in AGPRs.  Register plan and reasons: scripts/gen_ub_pg_asm.py, scripts/ubasm.py.\n"
               "#pragma once\n\nnamespace tpc {\n\n" % (H, n_iter, " (the compiler's loop: 708)" if H == 20 else "",
                                                      f"{NREG}..{H - 1}" if NA else "none"))
    out.append(f"constexpr int kUbAsmH{SUFFIX} = {H}, kUbAsmIterInstrs{SUFFIX} = {n_iter}, kUbAsmBatch{SUFFIX} = {BATCH};\n\n")
    if H == 20:
        out.append("struct UbAsmIn {\n"
                   "    const double* recs; const uint32_t* order; uint32_t* ticket; double* front; double* rear; int32_t* iters;\n"
                   "    uint32_t n_queue, max_iter;\n"
                   "    double gq0, gq1, grs0, grs1, grl0, grl1, lo0, lo1, hi0, hi1, s0, s1, xz0, xz1, geps;   // wave-uniform: MUST reach the asm in SGPRs (kernel arguments)\n"
                   "};\n")
    else:
        out.append("// (UbAsmIn: mpc_ub_pg_asm.h, included first by mpc_ub_asm.h)\n")
    out.append("// Runs the whole queue (every lane of the wavefront must be live).  Out: wave iterations, refill passes, TPC_MPC_FLAG bits.\n"
               "TPC_DEV void ub_pg_asm_run%s(const UbAsmIn& in, uint32_t& wave_iters, uint32_t& refills, uint32_t& flags) {\n" % SUFFIX
               + ("    uint32_t sdiag;\n" if DIAG else "") +
               "    uint64_t shave, sexh, swant, snew, stmp, sstop, sdonenow;\n"
               "    uint32_t scap, sleft, scnt, sfirst, sa, sb, scapf;\n"
               + ("    uint32_t scnext, scend, scbase;\n" if CHUNK else "") +
               "    asm volatile(\n")
    for l in body:
        out.append(f'        "{l}\\n"\n')
    outs = ['[sit] "=&s"(wave_iters)', '[srefills] "=&s"(refills)', '[sflags] "=&s"(flags)',
            '[shave] "=&s"(shave)', '[sexh] "=&s"(sexh)', '[swant] "=&s"(swant)', '[snew] "=&s"(snew)', '[stmp] "=&s"(stmp)',
            '[sstop] "=&s"(sstop)', '[sdonenow] "=&s"(sdonenow)', '[scap] "=&s"(scap)', '[sleft] "=&s"(sleft)', '[scnt] "=&s"(scnt)',
            '[sfirst] "=&s"(sfirst)', '[sa] "=&s"(sa)', '[sb] "=&s"(sb)', '[scapf] "=&s"(scapf)']
    if CHUNK:
        outs += ['[scnext] "=&s"(scnext)', '[scend] "=&s"(scend)', '[scbase] "=&s"(scbase)']
    ins = ['[precs] "s"(in.recs)', '[porder] "s"(in.order)', '[pticket] "s"(in.ticket)', '[pfront] "s"(in.front)', '[prear] "s"(in.rear)',
           '[piters] "s"(in.iters)', '[snq] "s"(in.n_queue)', '[smaxit] "s"(in.max_iter)',
           '[sgq0] "s"(in.gq0)', '[sgq1] "s"(in.gq1)', '[sgrs0] "s"(in.grs0)', '[sgrs1] "s"(in.grs1)', '[sgrl0] "s"(in.grl0)', '[sgrl1] "s"(in.grl1)',
           '[slo0] "s"(in.lo0)', '[slo1] "s"(in.lo1)', '[shi0] "s"(in.hi0)', '[shi1] "s"(in.hi1)', '[ss0] "s"(in.s0)', '[ss1] "s"(in.s1)',
           '[sxz0] "s"(in.xz0)', '[sxz1] "s"(in.xz1)', '[sgeps] "s"(in.geps)']
    clob = [f'"v{r}"' for r in range(NVGPR)] + [f'"a{r}"' for r in range(N_AGPR)] + ['"vcc"', '"scc"', '"memory"']
    if DIAG:
        outs.append('[sdiag] "=&s"(sdiag)')
        clob += ['"s96"', '"s97"', '"s98"', '"s99"']
    def wrap(items, ind):
        lines, cur = [], ""
        for it in items:
            if len(cur) + len(it) + 2 > 150:
                lines.append(cur.rstrip()); cur = ""
            cur += it + ", "
        lines.append(cur.rstrip().rstrip(","))
        return ("\n" + " " * ind).join(lines)
    out.append("        : " + wrap(outs, 10) + "\n")
    out.append("        : " + wrap(ins, 10) + "\n")
    out.append("        : " + wrap(clob, 10) + ");\n")
    if DIAG:
        out.append("    wave_iters = sdiag;   // DIAGNOSTIC BUILD %d: shader cycles of one section, not wave iterations\n" % DIAG)
    out.append("}\n\n}  // namespace tpc\n")
    return "".join(out)


if __name__ == "__main__":
    sys.stdout.write(gen())
