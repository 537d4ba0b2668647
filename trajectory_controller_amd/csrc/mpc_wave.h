// WAVE kernel: one 64-lane wavefront per MPC instance (the layout BASELINE.json's north_star
// describes).  Lane q owns decision variable q = i*I + j (i = horizon step, j = input).
//
//   set-up    every lane runs dlib's O(H) gradient recurrence on the unit vector e_q: that is column q
//             of K'QK, kept in the lane's VGPRs as "its row" (the Hessian Hd = K'QK + R is symmetric; the
//             R*u of the diagonal is added where the gradient is formed).  The lane's own diagonal entry
//             is dlib's Q_diag of its variable and the diagonal entries, R included, sum to dlib's lambda
//             (one wavefront sum), so the constructor's matrix recurrence is not run.  The linear term MM
//             is computed redundantly by all lanes; each keeps its own element.
//   loops     df_q = Hd[q,:].u + MM_q: one v_fmac_f64 per variable whose first operand is read through DPP
//             (row_newbcast: the control of lane K of the 16-lane row), after v_permlane16/32_swap has
//             given every row a copy of the other rows' controls -- no LDS, no wait.
//             Coordinate-descent iterations (iter < smo_iters) need the arg-max: the masked |df| with its
//             six lowest bits replaced by 63 - lane is a key whose wavefront maximum (butterfly DPP
//             exchanges) names value and winner at once, lowest index first like dlib's strict '>' scan
//             (mpc.h:292-308).  Projected-gradient iterations need only "is any free |df| >= eps": one
//             compare and a ballot.  No branch inside an iteration: a step is applied under its own stop
//             verdict as a select, and the loops look at the verdict once per block of iterations.
//
// Not bit-identical to dlib (the dot product sums in a different order; fused multiply-adds are
// used), but it takes dlib's decisions on dlib's quantities, so iteration counts agree and the
// outputs differ by ~1e-14 on the reference workload in fp64 (tests/test_parity_gpu.py).
// Iteration counts are wave-uniform: no divergence, no refill.  Supports I*H <= 64 with one variable per lane
// (wave_solve) and two inputs up to H = 64 with two per lane (wave2_solve).
#pragma once

#include <type_traits>
#include <utility>

#include "mpc_model.h"

namespace tpc {

template <class F, int... Is> TPC_DEV void static_for_w_impl(F&& f, std::integer_sequence<int, Is...>) {
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F> TPC_DEV void static_for_w(F&& f) { static_for_w_impl(f, std::make_integer_sequence<int, N>{}); }

template <int CTRL, int ROW_MASK = 0xf> TPC_DEV double dpp_mov(double old, double x) {
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(x), CTRL, ROW_MASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(x), CTRL, ROW_MASK, 0xf, false);
    return __hiloint2double(hi, lo);
}
template <int CTRL, int ROW_MASK = 0xf> TPC_DEV float dpp_mov(float old, float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(x), CTRL, ROW_MASK, 0xf, false));
}
TPC_DEV double read_lane(double x, int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), l),
                            __builtin_amdgcn_readlane(__double2loint(x), l));
}
TPC_DEV float read_lane(float x, int l) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), l));
}

// v_max on values that are already canonical (the operands come out of arithmetic or a lane move):
// the builtin would prepend a quieting v_max x, x to each operand it cannot prove canonical.
TPC_DEV double raw_max(double a, double b) { double r; asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
TPC_DEV float raw_max(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }

// lane-shifted copy inside a 16-lane row; lanes without a source read 0 (bound_ctrl), which is the
// neutral element here (the reduced values are >= 0)
template <int CTRL> TPC_DEV double dpp_shr0(double x) {
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(x), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(x), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
template <int CTRL> TPC_DEV float dpp_shr0(float x) {
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), CTRL, 0xf, 0xf, true));
}

// a0 += sum over even q, a1 += sum over odd q, of (x of lane q of the caller's 16-lane row) * k[q], q < CNT:
// v_fmac with a DPP row_newbcast first operand -- the broadcast and the multiply-add are one instruction.
// One asm block per row of controls: inline asm is opaque to the compiler's hazard recogniser, which
// otherwise pads every pair of separate asms with a nop; a VGPR written by the VALU needs two wait
// states before a DPP read, so the block opens with them (x may come straight out of an ALU op).
// Two accumulators: alternate v_fmacs depend on each other at distance 2, ~2 x 4.8 cycles of issue
// against 8.6 of latency.  S: stride of the entries in k (2 where a lane's row interleaves two inputs).
// The blocks are spelled by the preprocessor from ONE instruction template (TPC_FM_I): entry q of the row goes to
// accumulator q & 1, takes its multiplier from asm operand 3 + q and its broadcast lane either as the literal q
// (fmac_row) or from the immediate operand behind the multipliers (fmac_row8).
#define TPC_FM_I(SFX, ACC, KOP, BC) "v_fmac_" SFX "_dpp %" #ACC ", %2, %" #KOP " row_newbcast:" BC " row_mask:0xf bank_mask:0xf\n\t"
#define TPC_FM_1(S) TPC_FM_I(S, 0, 3, "0")
#define TPC_FM_2(S) TPC_FM_1(S) TPC_FM_I(S, 1, 4, "1")
#define TPC_FM_3(S) TPC_FM_2(S) TPC_FM_I(S, 0, 5, "2")
#define TPC_FM_4(S) TPC_FM_3(S) TPC_FM_I(S, 1, 6, "3")
#define TPC_FM_5(S) TPC_FM_4(S) TPC_FM_I(S, 0, 7, "4")
#define TPC_FM_6(S) TPC_FM_5(S) TPC_FM_I(S, 1, 8, "5")
#define TPC_FM_7(S) TPC_FM_6(S) TPC_FM_I(S, 0, 9, "6")
#define TPC_FM_8(S) TPC_FM_7(S) TPC_FM_I(S, 1, 10, "7")
#define TPC_FM_9(S) TPC_FM_8(S) TPC_FM_I(S, 0, 11, "8")
#define TPC_FM_10(S) TPC_FM_9(S) TPC_FM_I(S, 1, 12, "9")
#define TPC_FM_11(S) TPC_FM_10(S) TPC_FM_I(S, 0, 13, "10")
#define TPC_FM_12(S) TPC_FM_11(S) TPC_FM_I(S, 1, 14, "11")
#define TPC_FM_13(S) TPC_FM_12(S) TPC_FM_I(S, 0, 15, "12")
#define TPC_FM_14(S) TPC_FM_13(S) TPC_FM_I(S, 1, 16, "13")
#define TPC_FM_15(S) TPC_FM_14(S) TPC_FM_I(S, 0, 17, "14")
#define TPC_FM_16(S) TPC_FM_15(S) TPC_FM_I(S, 1, 18, "15")
#define TPC_FK_1 "v"(k[S * 0])
#define TPC_FK_2 TPC_FK_1, "v"(k[S * 1])
#define TPC_FK_3 TPC_FK_2, "v"(k[S * 2])
#define TPC_FK_4 TPC_FK_3, "v"(k[S * 3])
#define TPC_FK_5 TPC_FK_4, "v"(k[S * 4])
#define TPC_FK_6 TPC_FK_5, "v"(k[S * 5])
#define TPC_FK_7 TPC_FK_6, "v"(k[S * 6])
#define TPC_FK_8 TPC_FK_7, "v"(k[S * 7])
#define TPC_FK_9 TPC_FK_8, "v"(k[S * 8])
#define TPC_FK_10 TPC_FK_9, "v"(k[S * 9])
#define TPC_FK_11 TPC_FK_10, "v"(k[S * 10])
#define TPC_FK_12 TPC_FK_11, "v"(k[S * 11])
#define TPC_FK_13 TPC_FK_12, "v"(k[S * 12])
#define TPC_FK_14 TPC_FK_13, "v"(k[S * 13])
#define TPC_FK_15 TPC_FK_14, "v"(k[S * 14])
#define TPC_FK_16 TPC_FK_15, "v"(k[S * 15])
#define TPC_FMAC_ROW(N, SFX) if constexpr (CNT == N) asm volatile("s_nop 1\n\t" TPC_FM_##N(SFX) : "+v"(a0), "+v"(a1) : "v"(x), TPC_FK_##N)
template <int CNT, int S = 1> TPC_DEV void fmac_row(double& a0, double& a1, double x, const double* k) {
    static_assert(CNT >= 1 && CNT <= 16, "one 16-lane row");
    TPC_FMAC_ROW(1, "f64");
    TPC_FMAC_ROW(2, "f64");
    TPC_FMAC_ROW(3, "f64");
    TPC_FMAC_ROW(4, "f64");
    TPC_FMAC_ROW(5, "f64");
    TPC_FMAC_ROW(6, "f64");
    TPC_FMAC_ROW(7, "f64");
    TPC_FMAC_ROW(8, "f64");
    TPC_FMAC_ROW(9, "f64");
    TPC_FMAC_ROW(10, "f64");
    TPC_FMAC_ROW(11, "f64");
    TPC_FMAC_ROW(12, "f64");
    TPC_FMAC_ROW(13, "f64");
    TPC_FMAC_ROW(14, "f64");
    TPC_FMAC_ROW(15, "f64");
    TPC_FMAC_ROW(16, "f64");
}
template <int CNT, int S = 1> TPC_DEV void fmac_row(float& a0, float& a1, float x, const float* k) {
    static_assert(CNT >= 1 && CNT <= 16, "one 16-lane row");
    TPC_FMAC_ROW(1, "f32");
    TPC_FMAC_ROW(2, "f32");
    TPC_FMAC_ROW(3, "f32");
    TPC_FMAC_ROW(4, "f32");
    TPC_FMAC_ROW(5, "f32");
    TPC_FMAC_ROW(6, "f32");
    TPC_FMAC_ROW(7, "f32");
    TPC_FMAC_ROW(8, "f32");
    TPC_FMAC_ROW(9, "f32");
    TPC_FMAC_ROW(10, "f32");
    TPC_FMAC_ROW(11, "f32");
    TPC_FMAC_ROW(12, "f32");
    TPC_FMAC_ROW(13, "f32");
    TPC_FMAC_ROW(14, "f32");
    TPC_FMAC_ROW(15, "f32");
    TPC_FMAC_ROW(16, "f32");
}

// ... up to eight entries of a row starting at lane OFF (the broadcast lane is an immediate operand here)
#define TPC_F8_1(S) TPC_FM_I(S, 0, 3, "%c4")
#define TPC_F8_2(S) TPC_FM_I(S, 0, 3, "%c5") TPC_FM_I(S, 1, 4, "%c6")
#define TPC_F8_3(S) TPC_FM_I(S, 0, 3, "%c6") TPC_FM_I(S, 1, 4, "%c7") TPC_FM_I(S, 0, 5, "%c8")
#define TPC_F8_4(S) TPC_FM_I(S, 0, 3, "%c7") TPC_FM_I(S, 1, 4, "%c8") TPC_FM_I(S, 0, 5, "%c9") TPC_FM_I(S, 1, 6, "%c10")
#define TPC_F8_5(S) TPC_FM_I(S, 0, 3, "%c8") TPC_FM_I(S, 1, 4, "%c9") TPC_FM_I(S, 0, 5, "%c10") TPC_FM_I(S, 1, 6, "%c11") TPC_FM_I(S, 0, 7, "%c12")
#define TPC_F8_6(S) TPC_FM_I(S, 0, 3, "%c9") TPC_FM_I(S, 1, 4, "%c10") TPC_FM_I(S, 0, 5, "%c11") TPC_FM_I(S, 1, 6, "%c12") TPC_FM_I(S, 0, 7, "%c13") TPC_FM_I(S, 1, 8, "%c14")
#define TPC_F8_7(S) TPC_FM_I(S, 0, 3, "%c10") TPC_FM_I(S, 1, 4, "%c11") TPC_FM_I(S, 0, 5, "%c12") TPC_FM_I(S, 1, 6, "%c13") TPC_FM_I(S, 0, 7, "%c14") TPC_FM_I(S, 1, 8, "%c15") TPC_FM_I(S, 0, 9, "%c16")
#define TPC_F8_8(S) TPC_FM_I(S, 0, 3, "%c11") TPC_FM_I(S, 1, 4, "%c12") TPC_FM_I(S, 0, 5, "%c13") TPC_FM_I(S, 1, 6, "%c14") TPC_FM_I(S, 0, 7, "%c15") TPC_FM_I(S, 1, 8, "%c16") TPC_FM_I(S, 0, 9, "%c17") TPC_FM_I(S, 1, 10, "%c18")
#define TPC_FN_1 "n"(OFF + 0)
#define TPC_FN_2 TPC_FN_1, "n"(OFF + 1)
#define TPC_FN_3 TPC_FN_2, "n"(OFF + 2)
#define TPC_FN_4 TPC_FN_3, "n"(OFF + 3)
#define TPC_FN_5 TPC_FN_4, "n"(OFF + 4)
#define TPC_FN_6 TPC_FN_5, "n"(OFF + 5)
#define TPC_FN_7 TPC_FN_6, "n"(OFF + 6)
#define TPC_FN_8 TPC_FN_7, "n"(OFF + 7)
#define TPC_FMAC_ROW8(N, SFX) if constexpr (CNT == N) asm volatile("s_nop 1\n\t" TPC_F8_##N(SFX) : "+v"(a0), "+v"(a1) : "v"(x), TPC_FK_##N, TPC_FN_##N)
template <int CNT, int OFF, int S> TPC_DEV void fmac_row8(double& a0, double& a1, double x, const double* k) {
    static_assert(CNT >= 1 && CNT <= 8 && OFF + CNT <= 16, "part of one 16-lane row");
    TPC_FMAC_ROW8(1, "f64");
    TPC_FMAC_ROW8(2, "f64");
    TPC_FMAC_ROW8(3, "f64");
    TPC_FMAC_ROW8(4, "f64");
    TPC_FMAC_ROW8(5, "f64");
    TPC_FMAC_ROW8(6, "f64");
    TPC_FMAC_ROW8(7, "f64");
    TPC_FMAC_ROW8(8, "f64");
}
template <int CNT, int OFF, int S> TPC_DEV void fmac_row8(float& a0, float& a1, float x, const float* k) {
    static_assert(CNT >= 1 && CNT <= 8 && OFF + CNT <= 16, "part of one 16-lane row");
    TPC_FMAC_ROW8(1, "f32");
    TPC_FMAC_ROW8(2, "f32");
    TPC_FMAC_ROW8(3, "f32");
    TPC_FMAC_ROW8(4, "f32");
    TPC_FMAC_ROW8(5, "f32");
    TPC_FMAC_ROW8(6, "f32");
    TPC_FMAC_ROW8(7, "f32");
    TPC_FMAC_ROW8(8, "f32");
}

// v_permlane16_swap: (a, b) -> a' = rows (a0, b0, a2, b2), b' = rows (a1, b1, a3, b3);
// v_permlane32_swap: (a, b) -> a' = (a.lo32, b.lo32), b' = (a.hi32, b.hi32)   [measured on gfx950]
template <typename T> struct Swapped { T a, b; };
TPC_DEV Swapped<double> swap_rows16(double a, double b) {
    const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
    return {__hiloint2double((int)hi[0], (int)lo[0]), __hiloint2double((int)hi[1], (int)lo[1])};
}
TPC_DEV Swapped<float> swap_rows16(float a, float b) {
    const auto r = __builtin_amdgcn_permlane16_swap((unsigned)__float_as_int(a), (unsigned)__float_as_int(b), false, false);
    return {__int_as_float((int)r[0]), __int_as_float((int)r[1])};
}
TPC_DEV Swapped<double> swap_halves(double a, double b) {
    const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
    return {__hiloint2double((int)hi[0], (int)lo[0]), __hiloint2double((int)hi[1], (int)lo[1])};
}
TPC_DEV Swapped<float> swap_halves(float a, float b) {
    const auto r = __builtin_amdgcn_permlane32_swap((unsigned)__float_as_int(a), (unsigned)__float_as_int(b), false, false);
    return {__int_as_float((int)r[0]), __int_as_float((int)r[1])};
}

// per lane: bit of `mask` set ? a : b.  Written as asm so that a wave-uniform mask stays a select: the
// optimiser turns a select on a uniform condition into a branch around the work, and a branch is what
// the iteration loops are built to avoid.
TPC_DEV double lane_select(unsigned long long mask, double a, double b) {
    int lo, hi;
    asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(lo) : "v"(__double2loint(b)), "v"(__double2loint(a)), "s"(mask));
    asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(hi) : "v"(__double2hiint(b)), "v"(__double2hiint(a)), "s"(mask));
    return __hiloint2double(hi, lo);
}
TPC_DEV float lane_select(unsigned long long mask, float a, float b) {
    int r;
    asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(r) : "v"(__float_as_int(b)), "v"(__float_as_int(a)), "s"(mask));
    return __int_as_float(r);
}

// all ones if any bit of `mask` is set, else 0 -- on the scalar unit, as a lane mask for lane_select
// (written as `mask ? ~0 : 0` the compiler materialises it per lane in VGPRs)
TPC_DEV unsigned long long any_to_all(unsigned long long mask) {
    unsigned long long r;
    asm("s_cmp_lg_u64 %1, 0\n\ts_cselect_b64 %0, -1, 0" : "=s"(r) : "s"(mask) : "scc");
    return r;
}

// x >= 0 with its six lowest mantissa bits replaced by `tag` (0..63), and back (tag cleared)
TPC_DEV double pack_key_abs(double x, int tag) {   // of |x|
    return __hiloint2double(__double2hiint(x) & 0x7fffffff, (__double2loint(x) & ~63) | tag);
}
TPC_DEV double round_up_key(double x) {   // smallest value with a clear tag that is >= x (x > 0)
    const unsigned long long b = ((unsigned long long)__double_as_longlong(x) + 63ull) & ~63ull;
    return __longlong_as_double((long long)b);
}
TPC_DEV float round_up_key(float x) { return x; }
TPC_DEV double unpack_key(double x) { return __hiloint2double(__double2hiint(x), __double2loint(x) & ~63); }
// (fp32 keeps the two-step arg-max: six of its 23 mantissa bits is too much to give to the tag)

// max over lanes 0 .. N-1 (the other lanes hold 0), returned wave-uniform.  x >= 0; a NaN lane
// is ignored (v_max returns the other operand).  Only as many DPP steps as N needs: the reduction is
// a dependent chain, and it sits on the critical path of every coordinate-descent iteration.
template <int N, typename T> TPC_DEV T wave_max(T x) {
    if constexpr (N > 1) x = raw_max(x, dpp_shr0<0x111>(x));          // row_shr:1
    if constexpr (N > 2) x = raw_max(x, dpp_shr0<0x112>(x));          // row_shr:2
    if constexpr (N > 4) x = raw_max(x, dpp_shr0<0x114>(x));          // row_shr:4
    if constexpr (N > 8) x = raw_max(x, dpp_shr0<0x118>(x));          // row_shr:8  -> lane 15 of each row = row max
    if constexpr (N > 16) x = raw_max(x, dpp_mov<0x142, 0xa>(x, x));  // row_bcast:15 -> rows 1, 3
    if constexpr (N > 32) x = raw_max(x, dpp_mov<0x143, 0xc>(x, x));  // row_bcast:31 -> rows 2, 3; lane 63 = wave max
    constexpr int last = N > 32 ? 63 : (N > 16 ? 31 : (N > 8 ? 15 : (N > 4 ? 7 : (N > 2 ? 3 : (N > 1 ? 1 : 0)))));
    return read_lane(x, last);
}

// max over the lanes 0 .. N-1 of the first 16-lane row (N <= 16; lanes N .. 15 must hold values that cannot
// win), left in EVERY lane of the group: butterfly exchanges (xor 1, xor 2, half mirror, mirror) instead of
// shifts, so no broadcast or lane read follows.
template <int N> TPC_DEV double row_max_all(double x) {
    static_assert(N <= 16, "one row");
    auto step = [&](auto ctrl) {
        constexpr int c = decltype(ctrl)::value;
        const int lo = __builtin_amdgcn_mov_dpp(__double2loint(x), c, 0xf, 0xf, false);
        const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(x), c, 0xf, 0xf, false);
        x = raw_max(x, __hiloint2double(hi, lo));
    };
    if constexpr (N > 1) step(std::integral_constant<int, 0xB1>{});    // quad_perm [1,0,3,2]
    if constexpr (N > 2) step(std::integral_constant<int, 0x4E>{});    // quad_perm [2,3,0,1]
    if constexpr (N > 4) step(std::integral_constant<int, 0x141>{});   // row_half_mirror
    if constexpr (N > 8) step(std::integral_constant<int, 0x140>{});   // row_mirror
    return x;
}

// sum over the lanes 0 .. N-1 (the other lanes must hold 0), left in every lane of the rows in use:
// butterfly exchanges inside a row, row swaps across rows.
template <int N, typename T> TPC_DEV T wave_sum_all(T x) {
    auto step = [&](auto ctrl) {
        constexpr int c = decltype(ctrl)::value;
        if constexpr (sizeof(T) == 8) {
            const int lo = __builtin_amdgcn_mov_dpp(__double2loint((double)x), c, 0xf, 0xf, false);
            const int hi = __builtin_amdgcn_mov_dpp(__double2hiint((double)x), c, 0xf, 0xf, false);
            x = x + (T)__hiloint2double(hi, lo);
        } else {
            x = x + (T)__int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int((float)x), c, 0xf, 0xf, false));
        }
    };
    if constexpr (N > 1) step(std::integral_constant<int, 0xB1>{});    // quad_perm [1,0,3,2]
    if constexpr (N > 2) step(std::integral_constant<int, 0x4E>{});    // quad_perm [2,3,0,1]
    if constexpr (N > 4) step(std::integral_constant<int, 0x141>{});   // row_half_mirror
    if constexpr (N > 8) step(std::integral_constant<int, 0x140>{});   // row_mirror
    if constexpr (N > 16) { const Swapped<T> p = swap_rows16(x, x); x = p.a + p.b; }
    if constexpr (N > 32) { const Swapped<T> p = swap_halves(x, x); x = p.a + p.b; }
    return x;
}

template <typename T, int I, int H, class Args> struct WaveIO;
template <typename T, int I, int H> struct WaveIO<T, I, H, CompactArgs> {
    static TPC_DEV T init_u(const CompactArgs&, int64_t, int, int) { return (T)0; }
    static TPC_DEV T init_v(const CompactArgs&, int64_t, int, int) { return (T)0; }
    static TPC_DEV void write(const CompactArgs& g, int64_t k, bool active, int qi, int qj, T u, T, uint32_t it,
                              bool reporter = (threadIdx.x & 63) == 0) {
        if (active && qi == 0) {
            if (qj == 0) ((T*)g.front)[k] = u; else ((T*)g.rear)[k] = u;
        }
        if (g.iters && reporter) g.iters[k] = (int32_t)it;
    }
    static TPC_DEV void report(const CompactArgs&, uint32_t, uint32_t) {}
};
template <int I, int H> struct WaveIO<double, I, H, OneArgs> {
    static TPC_DEV double init_u(const OneArgs&, int64_t, int, int) { return 0.0; }
    static TPC_DEV double init_v(const OneArgs&, int64_t, int, int) { return 0.0; }
    static TPC_DEV void write(const OneArgs& g, int64_t, bool active, int qi, int qj, double u, double, uint32_t) {
        if (active && qi == 0)
            __hip_atomic_store(g.out + qj, (uint64_t)__double_as_longlong(u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    // what the batch entries report through flags_out / iters, for the one instance of tpc_mpc_solve_one: the host
    // waits for this word as it waits for the two outputs (tpc_mpc_one.hip), so the order of the three stores is free
    static TPC_DEV void report(const OneArgs& g, uint32_t flags, uint32_t it) {
        if ((threadIdx.x & 63) == 0)
            __hip_atomic_store(g.info, ((uint64_t)flags << 32) | (uint64_t)it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
};
template <typename T, int I, int H> struct WaveIO<T, I, H, GeneralArgs> {
    static TPC_DEV T init_u(const GeneralArgs& g, int64_t k, int qi, int qj) {
        if (!g.controls) return (T)0;
        const int src = (g.shift_controls && qi + 1 < H) ? qi + 1 : qi;   // mpc.h:231-232
        return ((const T*)g.controls)[(int64_t)(src * I + qj) * g.ld + k];
    }
    static TPC_DEV T init_v(const GeneralArgs& g, int64_t k, int qi, int qj) {
        return g.v ? ((const T*)g.v)[(int64_t)(qi * I + qj) * g.ld + k] : (T)0;
    }
    static TPC_DEV void write(const GeneralArgs& g, int64_t k, bool active, int qi, int qj, T u, T v, uint32_t it,
                              bool reporter = (threadIdx.x & 63) == 0) {
        if (active) {
            if (qi == 0) ((T*)g.u0)[(int64_t)qj * g.ld + k] = u;
            if (g.controls) ((T*)g.controls)[(int64_t)(qi * I + qj) * g.ld + k] = u;
            if (g.v) ((T*)g.v)[(int64_t)(qi * I + qj) * g.ld + k] = v;
        }
        if (g.iters && reporter) g.iters[k] = (int32_t)it;
    }
    static TPC_DEV void report(const GeneralArgs&, uint32_t, uint32_t) {}
};

// Row `slot` (= column, the Hessian is symmetric) of K'QK, i.e. of the Hessian Hd = K'QK + R WITHOUT its R
// (the diagonal's R*u is added where the gradient is formed: with R inside the row the compiler kept the
// row and the R-free values both alive, 60 registers at N = 40): dlib's gradient recurrences
// (mpc.h:275-283) applied to the unit vector e_(qi,qj) with MM = 0.  B*e is a column of B at step qi
// and zero elsewhere, so no control vector is materialised; the arithmetic is the generic one
// (x*1 = x, x + 0 = x exactly).  row[2*i + j] = (K'QK)[(i,j)][(qi,qj)].
// The lane's own diagonal entry is dlib's Q_diag of its variable: trans(B)*T_c*B with
// T_c = sum_k trans(A^k)*Q*A^k is exactly what the recurrence accumulates at (q, q) -- and the diagonal
// entries, R included, sum to dlib's lambda (its trace bound, mpc.h:116-123).  So the constructor's own
// O(H) matrix recurrence is not run at all; the values differ from dlib's by rounding only (a different
// association), like everything else in this family.
template <typename T, int I, int H, class Model>
TPC_DEV void hessian_row(const Model& m, bool active, int qi, int qj, T* row) {
    const T bq0 = active ? (qj == 0 ? m.B(0, 0) : m.B(0, I - 1)) : (T)0;
    const T bq1 = active ? (qj == 0 ? m.B(1, 0) : m.B(1, I - 1)) : (T)0;
    T m0 = (T)0, m1 = (T)0;
#pragma unroll
    for (int i = 0; i < H; ++i) {
        const T s0 = (i == qi) ? bq0 : (T)0, s1 = (i == qi) ? bq1 : (T)0;
        const T n0 = (m.A(0, 0) * m0 + m.A(0, 1) * m1) + s0;
        const T n1 = (m.A(1, 0) * m0 + m.A(1, 1) * m1) + s1;
        m0 = n0; m1 = n1;
        row[2 * i] = m0; row[2 * i + 1] = m1;
    }
    T n0 = row[2 * (H - 1)] * m.Q(0), n1 = row[2 * (H - 1) + 1] * m.Q(1);
#pragma unroll
    for (int i = H - 1; i >= 0; --i) {
        if (i < H - 1) {
            const T t0 = row[2 * i] * m.Q(0) + (m.A(0, 0) * n0 + m.A(1, 0) * n1);
            const T t1 = row[2 * i + 1] * m.Q(1) + (m.A(0, 1) * n0 + m.A(1, 1) * n1);
            n0 = t0; n1 = t1;
        }
#pragma unroll
        for (int j = 0; j < I; ++j) {
            const T btn = m.B(0, j) * n0 + m.B(1, j) * n1;
            row[2 * i + j] = btn;
        }
    }
}

// a[idx] for a per-lane idx < LEN with the array in registers: a binary tree of selects on the bits of idx
// (LEN - 1 selects under log2(LEN) lane masks; a chain of "idx == k ? a[k]" needs LEN masks, and keeping 2H of
// them alive cost the N = 40 kernel 60 registers)
template <typename T, int LEN> TPC_DEV T pick_own(const T* a, int idx) {
    T lvl[LEN];
#pragma unroll
    for (int k = 0; k < LEN; ++k) lvl[k] = a[k];
    int n = LEN;
#pragma unroll
    for (int bit = 0; (1 << bit) < LEN; ++bit) {
        const bool odd = (idx >> bit) & 1;
        const int half = (n + 1) / 2;
#pragma unroll
        for (int k = 0; k < half; ++k) lvl[k] = (2 * k + 1 < n) ? (odd ? lvl[2 * k + 1] : lvl[2 * k]) : lvl[2 * k];
        n = half;
    }
    return lvl[0];
}

struct NoHook { TPC_DEV void operator()() const {} };

// What wave_solve's set-up derives from the MODEL alone (v and the parameters; not the targets): this lane's Hessian
// row, Q_diag, lambda and the constants made from them.  A resident wavefront (tpc_mpc_one.hip) keeps one across
// requests and sets `hit` when a request repeats the previous one's model, which skips ~300 of the ~450
// instructions before the first iteration; batch kernels use a throw-away one (hit = false folds away).
template <typename T, int H> struct WaveKeep {
    bool hit = false;   // wave-uniform
    T row[2 * H];
    T a, c;             // compact model: step * v, step * v / wheelbase
    T my_qd, lambda, my_rqd, inv_lambda, beta;
};
struct NoKeep {};

// One instance solved by the calling wavefront (all 64 lanes must call it together).
// s_w: 2*H values of LDS private to the wavefront, 16-byte aligned.
// before_loops(): called once, after the set-up has consumed everything it loads and before the iteration loops --
// the place for a memory operation whose result is wanted after the solve (the work queue's next ticket: the
// wait counter is in-order, so issued any earlier it is waited for together with the model's own loads).
template <typename T, int I, int H, class Model, class Args, class Hook = NoHook, class Keep = NoKeep>
TPC_DEV void wave_solve(const Args& g, const Knobs& kn, int64_t k, T* s_w, Hook before_loops = Hook{}, Keep* keep = nullptr) {
    constexpr int N = I * H;
    static_assert(N <= kWave, "WAVE kernel: one variable per lane");
    const int lane = threadIdx.x & (kWave - 1);
    const bool active = lane < N;
    const int qi = active ? lane / I : 0, qj = active ? lane % I : 0;
    const int slot = 2 * qi + qj;

    constexpr bool KEEP = !std::is_same<Keep, NoKeep>::value;
    WaveKeep<T, H> scratch_keep;
    WaveKeep<T, H>* kp = &scratch_keep;
    if constexpr (KEEP) kp = keep;
    const bool hit = KEEP && kp->hit;

    Model m;
    if constexpr (KEEP && std::is_same<Model, CompactModel<T>>::value) {
        m.load_targets_and_uniforms(g);
        if (hit) { m.a = kp->a; m.c = kp->c; }
        else { m.load(g, k); kp->a = m.a; kp->c = m.c; }
    } else {
        m.load(g, k);   // every lane reads the same instance: broadcast loads
    }
    const bool nonfinite = m.nonfinite();
    const bool badmodel = m.invalid();   // dlib's requires clause broken: return the start point

    // ---- prologue: this lane's Hessian row, Q_diag, linear-term element, lambda
    T* const row = kp->row;
    T my_g = (T)0;
    if (!hit) {
        hessian_row<T, I, H>(m, active, qi, qj, row);
        kp->my_qd = active ? pick_own<T, 2 * H>(row, slot) : (T)0;   // dlib's Q_diag of this lane's variable (mpc.h:118-121)
    }
    const T my_qd = kp->my_qd;
    // the entries of the row in variable order: the row itself for two inputs, every second slot of it for one
    T kq_one[I == 2 ? 1 : N];
    const T* kq = row;
    if constexpr (I != 2) {
#pragma unroll
        for (int q = 0; q < N; ++q) kq_one[q] = row[2 * (q / I) + (q % I)];
        kq = kq_one;
    }
    const T my_r = active ? m.R(qj) : (T)0;
    if (!hit) kp->lambda = wave_sum_all<N>(my_qd + my_r);   // trace of the Hessian (idle lanes hold 0 + 0)
    const T lambda = kp->lambda;
    // every lane computes the same linear term.  The compact model needs no intermediates (its own
    // linear_term: one target for all steps); a short general horizon keeps them in registers; a long
    // one parks them -- identical in all lanes -- in one small LDS vector instead of 2H registers per lane
    auto take_g = [&](int q, T val) { if (q == slot) my_g = val; };
    if constexpr (std::is_same<Model, CompactModel<T>>::value) {
        linear_term<T, I, H>(m, (T*)nullptr, take_g);
    } else if constexpr (H <= 10) {
        T w[2 * H];
        linear_term<T, I, H>(m, w, take_g);
    } else {
        linear_term_fn<T, I, H>(m, [&](int q, T val) { s_w[q] = val; }, [&](int q) { return s_w[q]; }, take_g);
    }
    if (!active) my_g = (T)0;   // an idle lane shadows variable 0 through the prologue; from here on it is all zeros
    const T lo = m.lo(qj), hi = m.hi(qj);
    const T eps = (T)kn.eps;
    // the coordinate step divides by Q_diag (mpc.h:325): one correctly rounded reciprocal per lane
    // here instead of a ~13-instruction dependent division chain in every coordinate-descent
    // iteration (the product differs from the quotient by an ulp at most: within this family's
    // tolerance, like its FMA dot product)
    if (!hit) {
        kp->my_rqd = (T)1 / my_qd;
        kp->inv_lambda = (T)1.0 / lambda;                 // mpc.h:342
        const T sq = tsqrt(lambda);
        kp->beta = (sq - (T)1) / (sq + (T)1);             // mpc.h:343
    }
    const T my_rqd = kp->my_rqd, inv_lambda = kp->inv_lambda, beta = kp->beta;

    T u = active ? WaveIO<T, I, H, Args>::init_u(g, k, qi, qj) : (T)0;
    T v = active ? WaveIO<T, I, H, Args>::init_v(g, k, qi, qj) : (T)0;

    // The masked |df| of mpc.h:298-299 comes in two forms.  Exact: dlib's compares, whose results travel
    // VALU -> SALU -> VALU twice per iteration, on the critical path.  Arithmetic (MASK): g_lo =
    // (u - lo) * 2^600 and g_hi = (hi - u) * 2^600 are 0 on the bound and huge off it, depend on u
    // only (so they are computed beside the dot product), and |max(min(df, g_lo), -g_hi)| is the masked
    // |df| -- two dependent instructions behind df.  Same value wherever no df can be NaN and the
    // controls stay inside bounds that straddle zero, which the model's screen proves for this
    // instance (one decision per wavefront: everything it looks at is wave-uniform).
    bool mask_ok = false;
    if constexpr (Model::kFastStop) {
        const T mm_max = wave_max<N>(active ? tabs(my_g) : (T)0);
        const bool start_inside = u >= lo && u <= hi;   // a caller's warm start may lie outside
        const bool term_nan = my_g != my_g;             // wave_max ignores a NaN lane: look for one explicitly
        mask_ok = m.fast_stop_ok(mm_max, eps, lambda, H) && __ballot(active && (!start_inside || term_nan)) == 0ull;
    }
    constexpr T kHuge = (T)(sizeof(T) == 8 ? 0x1p600 : 0x1p100);
    const T nlo_h = -(lo * kHuge), hi_h = hi * kHuge;
    T huge_r = kHuge;                    // in a register: as a literal it forces the two-operand v_fmac
    asm volatile("" : "+v"(huge_r));     // form, which costs a copy of the addend per use

    // df_q = Hd[q,:].u + g_q: one v_fmac with a DPP row_newbcast operand per variable (the control of
    // lane K of the row, times this lane's Hessian entry, in one instruction: no move, no LDS, no wait).
    // A DPP read reaches the caller's 16-lane row only, so with more than 16 variables every row first
    // gets a copy of the other rows' controls: X[r] = "the controls of row r, in every row", made by
    // v_permlane16_swap (rows 0<->1, 2<->3) and, past 32 variables, v_permlane32_swap (halves).
    auto gradient_of = [&](T uu) -> T {
        // two accumulators: with a dependent v_fmac every second instruction the chain never waits
        // (8.6 cycles of latency against 2 x 4.8 of issue), and the linear term seeds one of them
        T a0 = my_g, a1 = my_r * uu;   // (the Hessian's R, on the diagonal)
        T x[4] = {uu, uu, uu, uu};
        if constexpr (N > 16) {
            const Swapped<T> p = swap_rows16(uu, uu);        // p.a = rows (0,0,2,2), p.b = rows (1,1,3,3)
            x[0] = p.a; x[1] = p.b;
            if constexpr (N > 32) {
                const Swapped<T> e = swap_halves(p.a, p.a);  // e.a = row 0 everywhere, e.b = row 2 everywhere
                const Swapped<T> o = swap_halves(p.b, p.b);
                x[0] = e.a; x[2] = e.b; x[1] = o.a; x[3] = o.b;
            }
        }
        static_for_w<(N + 15) / 16>([&](auto rc) {
            constexpr int r = decltype(rc)::value;
            constexpr int cnt = N - 16 * r < 16 ? N - 16 * r : 16;
            fmac_row<cnt>(a0, a1, x[r], kq + 16 * r);
        });
        return a0 + a1;
    };

    constexpr int kUnrollCd = 5, kUnrollPg = 8;
    uint32_t iter = 0;
    bool capped = true;
    // Two loops, one per phase; both kinds of step are computed SPECULATIVELY beside the stop test they do
    // not depend on (one wave's fp64 instructions issue every ~5 cycles when independent and every ~9 when
    // each needs the previous result).
    auto run = [&](auto mask_tag) {
        constexpr bool MASK = decltype(mask_tag)::value;
        // the masked df, signed (callers take the magnitude where it is free: a compare's source modifier,
        // the key's sign bit)
        auto masked = [&](T uu, T df) -> T {
            if constexpr (MASK) {
                // (an idle lane needs no select here: its row, linear term and controls are 0, so df = 0,
                // and with bounds that straddle zero the expression below is 0)
                const T g_lo = tfma(uu, huge_r, nlo_h), g_hi = tfma(uu, -huge_r, hi_h);
                return tmax(tmin(df, g_lo), -g_hi);
            } else {
                const bool blocked = (uu <= lo && df > (T)0) || (uu >= hi && df < (T)0);   // mpc.h:298-299
                return (active && !blocked) ? df : (T)0;
            }
        };
        // A branch instruction costs a wavefront ~70 cycles, taken or not (measured: the same loop without
        // its exit branch), against ~5 for an ALU instruction -- a quarter of an iteration at 8 variables.
        // So an iteration is BRANCH-FREE: its step is applied under the verdict of its own stop test as a
        // select, and the iteration counter advances by that verdict.  Once the test says stop, the state
        // no longer changes, so the following iterations reproduce the same verdict and do nothing; the
        // loops look at the verdict once per block of kUnroll iterations (a few idle iterations at the end
        // of a solve against a branch in every one).
        const bool qd_nz = my_qd != (T)0;
        const T my_rqd0 = qd_nz ? my_rqd : (T)0;
        unsigned long long go_mask = ~0ull;
        bool go_lane = true, last_take = false;
        int cnt = 0;                          // iterations this lane has seen counted (all lanes alike)
        // eps rounded up to the key grid (a multiple of 64 ulp): key >= eps_up  <=>  key with its tag cleared >= eps
        const T eps_up = round_up_key(eps);
        auto cd_step = [&]() {
            const T df = gradient_of(u);
            const T cs = masked(u, df);
            // mpc.h:325-326, every lane its own (speculatively: only the winner's is used): -(df - Qd*u)/Qd is
            // the Newton step u - df/Qd.  Where the screen ran it is ONE fused multiply-add with the lane's
            // reciprocal -- which is 0 for a zero Q_diag, so that lane "updates" to the value it has
            // (mpc.h:322: the update is skipped, the iteration still counts); elsewhere dlib's expression
            // and an explicit select (a warm start may lie outside the bounds, where clamp(u) != u).
            T nu;
            if constexpr (MASK) nu = tmax(tmin(tfma(-my_rqd0, df, u), hi), lo);
            else {
                nu = put_in_range(lo, hi, -(df - my_qd * u) * my_rqd);
                nu = qd_nz ? nu : u;
            }
            asm volatile("" : "+v"(nu));   // computed HERE, beside the reduction (the optimiser would sink it behind the winner's mask)
            if constexpr (sizeof(T) == 8) {
                // arg-max and max in ONE reduction: the low six bits of the masked |df| are replaced by
                // 63 - lane, so the largest key is unique, belongs to the lowest lane among (near-)equal
                // values as dlib's strict '>' scan would pick, and differs from the true maximum by
                // < 2^-46 relative -- inside what this family's FMA dot product already differs by.
                // (an infinite |df| -- possible only where the screen did not run -- would turn into a NaN
                // under the tag and drop out of the maximum: it competes as the largest finite number)
                const T cf = (!MASK && tabs(cs) == (T)__builtin_inf()) ? (T)1.7976931348623157e308 : cs;
                const T key = pack_key_abs(cf, 63 - lane);
                // The winner and the verdict without leaving the vector unit (every hop VALU -> SALU -> VALU
                // costs ~20 cycles of a chain that has nothing else to do): the maximum reaches every lane
                // (butterfly for one row; a lane read otherwise), and "this lane holds it AND it is >= eps" is
                // ONE compare against max(maximum, eps rounded up to the key grid).
                T mk;
                if constexpr (N <= 16) mk = row_max_all<N>(key);
                else mk = wave_max<N>(key);
                const T thr = raw_max(mk, eps_up);
                const bool take = key >= thr;
                go_lane = mk >= eps_up;                                          // mpc.h:310-311 (all lanes alike)
                last_take = take;
                u = take ? nu : u;
            } else {
                const T c = tabs(cs);
                const T mx = wave_max<N>(c);
                go_lane = !(mx < eps);
                const unsigned long long hit = __ballot(c == mx);
                const bool take = go_lane && lane == __ffsll((long long)hit) - 1;   // lowest index wins
                last_take = take;
                u = take ? nu : u;
            }
            cnt += go_lane ? 1 : 0;
        };
        auto pg_step = [&]() {
            const T df = gradient_of(u);
            const T cs = masked(u, df);
            const T v_new = clamp3(tfma(-inv_lambda, df, u), lo, hi);            // fused, like the dot product above
            const T u_new = clamp3(tfma(beta, v_new - v, v_new), lo, hi);
            go_mask = __ballot(tabs(cs) >= eps);                                 // mpc.h:310-311: 0 = stop
            const unsigned long long go_all = any_to_all(go_mask);
            v = lane_select(go_all, v_new, v);
            u = lane_select(go_all, u_new, u);
            iter += go_mask != 0ull ? 1u : 0u;
        };
        // ---- coordinate descent on the arg-max (mpc.h:319-335)
        const uint32_t cd_end = kn.smo_iters < kn.max_iter ? kn.smo_iters : kn.max_iter;
        const uint32_t cd_left = cd_end > iter ? cd_end - iter : 0u;
        for (uint32_t blk = cd_left / kUnrollCd; blk;) {
#pragma unroll
            for (int r = 0; r < kUnrollCd; ++r) cd_step();
            go_mask = __ballot(go_lane);
            blk = go_mask != 0ull ? blk - 1 : 0u;          // one branch per block: the loop's own
        }
        for (uint32_t rest = go_mask != 0ull ? cd_left % kUnrollCd : 0u; rest; --rest) cd_step();
        go_mask = __ballot(go_lane);
        iter += (uint32_t)__builtin_amdgcn_readfirstlane(cnt);
        if (go_mask == 0ull) { capped = false; return; }
        if (iter == kn.smo_iters && __ballot(last_take && qd_nz) != 0ull) v = u;   // mpc.h:330-334: the last CD iteration, unless it was skipped
        // ---- accelerated projected gradient (mpc.h:336-345); stop test without a reduction
        const uint32_t pg_left = kn.max_iter > iter ? kn.max_iter - iter : 0u;
        for (uint32_t blk = pg_left / kUnrollPg; blk;) {
#pragma unroll
            for (int r = 0; r < kUnrollPg; ++r) pg_step();
            blk = go_mask != 0ull ? blk - 1 : 0u;
        }
        for (uint32_t rest = go_mask != 0ull ? pg_left % kUnrollPg : 0u; rest; --rest) pg_step();
        if (go_mask == 0ull) capped = false;
    };
    before_loops();
    if ((Model::kScreen && nonfinite) || badmodel) capped = false;
    else if (mask_ok) run(std::true_type{});
    else run(std::false_type{});
    uint32_t f = 0;   // (all lanes alike)
    if (nonfinite) f |= 0x1u;
    if (badmodel) f |= 0x4u;
    if (capped) f |= 0x2u;
    WaveIO<T, I, H, Args>::report(g, f, iter);
    WaveIO<T, I, H, Args>::write(g, k, active, qi, qj, u, v, iter);
    if (g.flags) raise_flags(g.flags, f);
}

// ---- two instances per wavefront ----------------------------------------------------------------------------
// One wavefront's instruction issue fills a SIMD (one instruction per 4 cycles), so past two wavefronts per SIMD a
// WAVE batch runs at the instruction rate of one instance per SIMD -- with 20 of 64 lanes in use at N = 10.  Up to
// 32 variables an instance fits one half of the wavefront (two 16-lane rows), and everything the iteration moves
// between lanes already stays inside a half: the DPP row broadcast, v_permlane16_swap (rows 0<->1, 2<->3), the
// butterfly sums.  wave_pair_solve runs instance k0 in lanes 0..31 and k1 in lanes 32..63 through the same
// instructions: same arithmetic per instance as wave_solve (the row-local and half-local reductions associate
// exactly as the one-instance ones do for N <= 32), every verdict per half.  The two instances stop at different
// iterations; the one that stopped keeps its state (a stopped state reproduces its stop verdict), so the pair costs
// max(iterations) -- the work queue hands out neighbours of the longest-first order, which are alike.
// k1 < 0: no second instance.  fp64 only (the fp32 arg-max is a two-step wavefront reduction).
TPC_DEV unsigned long long halves_any_to_all(unsigned long long mask) {   // per 32-lane half: any bit set -> all ones
    unsigned lo, hi;
    asm("s_cmp_lg_u32 %1, 0\n\ts_cselect_b32 %0, -1, 0" : "=s"(lo) : "s"((unsigned)mask) : "scc");
    asm("s_cmp_lg_u32 %1, 0\n\ts_cselect_b32 %0, -1, 0" : "=s"(hi) : "s"((unsigned)(mask >> 32)) : "scc");
    return ((unsigned long long)hi << 32) | lo;
}
// per 16-lane row: any bit set -> all ones (carry-free field arithmetic on the scalar unit: no select, no branch)
TPC_DEV unsigned long long rows_any_to_all(unsigned long long mask) {
    auto spread = [](unsigned w) {
        const unsigned top = (((w & 0x7fff7fffu) + 0x7fff7fffu) | w) & 0x80008000u;   // bit 15 of a field: field != 0
        return (top >> 15) * 0xffffu;
    };
    return ((unsigned long long)spread((unsigned)(mask >> 32)) << 32) | spread((unsigned)mask);
}
template <int G> TPC_DEV unsigned long long groups_any_to_all(unsigned long long mask) {
    if constexpr (G == 2) return halves_any_to_all(mask);
    else return rows_any_to_all(mask);
}
TPC_DEV uint32_t add_lane_bit(uint32_t x, unsigned long long mask) {   // x + (bit of this lane in mask): one v_addc
    uint32_t r;
    unsigned long long carry_out;
    asm("v_addc_co_u32_e64 %0, %1, %2, 0, %3" : "=v"(r), "=s"(carry_out) : "v"(x), "s"(mask));
    return r;
}
// max over the lanes of the caller's 32-lane half, left in every lane of it (x >= 0; lanes past N hold 0)
template <int N> TPC_DEV double half_max_all(double x) {
    if constexpr (N <= 16) return row_max_all<N>(x);
    else {
        x = row_max_all<16>(x);
        const Swapped<double> p = swap_rows16(x, x);
        return raw_max(p.a, p.b);
    }
}

// G = 4: the same with one instance per 16-lane row (at most 16 variables: N = 4 and 5 with two inputs), ks[0..3].
template <typename T, int I, int H, int G, class Model, class Args, class Hook = NoHook>
TPC_DEV void wave_pair_solve(const Args& g, const Knobs& kn, const int64_t (&ks)[G], T* s_w, Hook before_loops = Hook{}) {
    static_assert(sizeof(T) == 8, "fp64 only");
    static_assert(G == 2 || G == 4, "a half or a row per instance");
    constexpr int N = I * H, L = kWave / G;
    static_assert(N <= L, "one instance per group of lanes");
    const int lane = threadIdx.x & (kWave - 1);
    const int half = lane / L, ll = lane % L;               // (half: the lane's group -- a 32-lane half or a 16-lane row)
    int64_t k_mine = ks[0];
#pragma unroll
    for (int j = 1; j < G; ++j) k_mine = half == j ? ks[j] : k_mine;
    const bool present = k_mine >= 0;
    const int64_t k = present ? k_mine : ks[0];             // (a missing instance shadows the first: valid addresses)
    const bool owns = ll < N && present;                     // this lane holds a variable of an instance
    const int qi = owns ? ll / I : 0, qj = owns ? ll % I : 0;
    const int slot = 2 * qi + qj;
    T* lt = s_w + half * 2 * H;

    Model m;
    m.load(g, k);
    const bool nonfinite = m.nonfinite();
    const bool badmodel = m.invalid();
    const bool alive = present && !((Model::kScreen && nonfinite) || badmodel);   // the half iterates
    const bool active = owns && alive;
    const unsigned long long alive_mask = __ballot(alive);

    T row[2 * H];
    T my_qd = (T)0, my_g = (T)0;
    hessian_row<T, I, H>(m, active, qi, qj, row);
    if (!active) {   // a half that does not iterate (or an idle lane) must not feed NaNs of a broken model into its verdicts
#pragma unroll
        for (int q = 0; q < 2 * H; ++q) row[q] = (T)0;
    }
    my_qd = active ? pick_own<T, 2 * H>(row, slot) : (T)0;
    T kq_one[I == 2 ? 1 : N];
    const T* kq = row;
    if constexpr (I != 2) {
#pragma unroll
        for (int q = 0; q < N; ++q) kq_one[q] = row[2 * (q / I) + (q % I)];
        kq = kq_one;
    }
    const T my_r = active ? m.R(qj) : (T)0;
    const T lambda = wave_sum_all<N>(my_qd + my_r);   // (N <= 32: the butterfly stays inside the half)
    auto take_g = [&](int q, T val) { if (q == slot) my_g = val; };
    if constexpr (std::is_same<Model, CompactModel<T>>::value) {
        linear_term<T, I, H>(m, (T*)nullptr, take_g);
    } else if constexpr (H <= 10) {
        T w[2 * H];
        linear_term<T, I, H>(m, w, take_g);
    } else {
        linear_term_fn<T, I, H>(m, [&](int q, T val) { lt[q] = val; }, [&](int q) { return lt[q]; }, take_g);
    }
    if (!active) my_g = (T)0;
    const T lo = m.lo(qj), hi = m.hi(qj);
    const T eps = (T)kn.eps;
    const T my_rqd = (T)1 / my_qd;
    const T inv_lambda = (T)1.0 / lambda;
    const T sq = tsqrt(lambda);
    const T beta = (sq - (T)1) / (sq + (T)1);

    T u = owns ? WaveIO<T, I, H, Args>::init_u(g, k, qi, qj) : (T)0;
    T v = owns ? WaveIO<T, I, H, Args>::init_v(g, k, qi, qj) : (T)0;

    bool mask_ok = false;
    if constexpr (Model::kFastStop) {
        const T mm_max = half_max_all<N>(active ? tabs(my_g) : (T)0);
        const bool start_inside = u >= lo && u <= hi;
        const bool term_nan = my_g != my_g;
        const bool fine = m.fast_stop_ok(mm_max, eps, lambda, H) && start_inside && !term_nan;
        mask_ok = __ballot(active && !fine) == 0ull;     // one code path for the wavefront: both halves must qualify
    }
    constexpr T kHuge = (T)0x1p600;
    const T nlo_h = -(lo * kHuge), hi_h = hi * kHuge;
    T huge_r = kHuge;
    asm volatile("" : "+v"(huge_r));

    auto gradient_of = [&](T uu) -> T {
        T a0 = my_g, a1 = my_r * uu;
        T x[2] = {uu, uu};
        if constexpr (N > 16) {
            const Swapped<T> p = swap_rows16(uu, uu);   // rows (0,0,2,2) and (1,1,3,3): each half sees its own two rows
            x[0] = p.a; x[1] = p.b;
        }
        static_for_w<(N + 15) / 16>([&](auto rc) {
            constexpr int r = decltype(rc)::value;
            constexpr int cnt = N - 16 * r < 16 ? N - 16 * r : 16;
            fmac_row<cnt>(a0, a1, x[r], kq + 16 * r);
        });
        return a0 + a1;
    };

    constexpr int kUnrollCd = 5, kUnrollPg = 8;
    uint32_t iter_l = 0;              // iterations of this lane's instance (alike within a half)
    unsigned long long still = 0ull;  // halves that were still going when the loops ended (= ran into the cap)
    auto run = [&](auto mask_tag) {
        constexpr bool MASK = decltype(mask_tag)::value;
        auto masked = [&](T uu, T df) -> T {
            if constexpr (MASK) {
                const T g_lo = tfma(uu, huge_r, nlo_h), g_hi = tfma(uu, -huge_r, hi_h);
                return tmax(tmin(df, g_lo), -g_hi);
            } else {
                const bool blocked = (uu <= lo && df > (T)0) || (uu >= hi && df < (T)0);   // mpc.h:298-299
                return (active && !blocked) ? df : (T)0;
            }
        };
        const bool qd_nz = my_qd != (T)0;
        const T my_rqd0 = qd_nz ? my_rqd : (T)0;
        unsigned long long go_mask = alive_mask;
        unsigned long long pg_alive = alive_mask;   // halves that enter the projected-gradient phase (a half that stopped
                                                    // during coordinate descent is finished: wave_solve returns there)
        bool go_lane = alive, last_take = false;
        uint32_t cnt = 0;
        const T eps_up = round_up_key(eps);
        auto cd_step = [&]() {
            const T df = gradient_of(u);
            const T cs = masked(u, df);
            T nu;
            if constexpr (MASK) nu = tmax(tmin(tfma(-my_rqd0, df, u), hi), lo);
            else {
                nu = put_in_range(lo, hi, -(df - my_qd * u) * my_rqd);
                nu = qd_nz ? nu : u;
            }
            asm volatile("" : "+v"(nu));
            const T cf = (!MASK && tabs(cs) == (T)__builtin_inf()) ? (T)1.7976931348623157e308 : cs;
            const T key = pack_key_abs(cf, 63 - lane);      // (the tag orders the lanes of a half like those of a wavefront)
            const T mk = half_max_all<N>(key);
            const T thr = raw_max(mk, eps_up);
            const bool take = alive && key >= thr;
            go_lane = alive && mk >= eps_up;                // mpc.h:310-311 (all lanes of a half alike)
            last_take = take;
            u = take ? nu : u;
            cnt += go_lane ? 1u : 0u;
        };
        auto pg_step = [&]() {
            const T df = gradient_of(u);
            const T cs = masked(u, df);
            const T v_new = clamp3(tfma(-inv_lambda, df, u), lo, hi);
            const T u_new = clamp3(tfma(beta, v_new - v, v_new), lo, hi);
            go_mask = __ballot(tabs(cs) >= eps) & pg_alive;                      // mpc.h:310-311, per half below
            const unsigned long long go_all = groups_any_to_all<G>(go_mask);
            v = lane_select(go_all, v_new, v);
            u = lane_select(go_all, u_new, u);
            iter_l = add_lane_bit(iter_l, go_all);
            still = go_all;
        };
        // ---- coordinate descent (mpc.h:319-335); the iteration index is common to both halves
        const uint32_t cd_end = kn.smo_iters < kn.max_iter ? kn.smo_iters : kn.max_iter;
        for (uint32_t blk = cd_end / kUnrollCd; blk;) {
#pragma unroll
            for (int r = 0; r < kUnrollCd; ++r) cd_step();
            go_mask = __ballot(go_lane);
            blk = go_mask != 0ull ? blk - 1 : 0u;
        }
        for (uint32_t rest = go_mask != 0ull ? cd_end % kUnrollCd : 0u; rest; --rest) cd_step();
        go_mask = __ballot(go_lane);
        iter_l = cnt;
        still = groups_any_to_all<G>(go_mask);
        if (go_mask == 0ull) return;
        {   // mpc.h:330-334, per half: the last CD iteration seeds v unless it was skipped
            const unsigned long long seeded = groups_any_to_all<G>(__ballot(last_take && qd_nz));
            const bool mine = ((seeded >> lane) & 1ull) != 0ull;
            if (mine && iter_l == kn.smo_iters) v = u;
        }
        // ---- accelerated projected gradient (mpc.h:336-345)
        const uint32_t pg_left = kn.max_iter > cd_end ? kn.max_iter - cd_end : 0u;
        pg_alive = still & alive_mask;
        go_mask = pg_alive;
        for (uint32_t blk = pg_left / kUnrollPg; blk;) {
#pragma unroll
            for (int r = 0; r < kUnrollPg; ++r) pg_step();
            blk = go_mask != 0ull ? blk - 1 : 0u;
        }
        for (uint32_t rest = go_mask != 0ull ? pg_left % kUnrollPg : 0u; rest; --rest) pg_step();
    };
    before_loops();
    if (alive_mask != 0ull) {
        if (mask_ok) run(std::true_type{});
        else run(std::false_type{});
    }
    WaveIO<T, I, H, Args>::write(g, k, owns, qi, qj, u, v, iter_l, ll == 0 && present);
    if (g.flags) {
        const bool capped = ((still >> lane) & 1ull) != 0ull;
        uint32_t f = 0;
        if (present && nonfinite) f |= 0x1u;
        if (present && badmodel) f |= 0x4u;
        if (capped) f |= 0x2u;
        raise_flags(g.flags, f);
    }
}

// kWavesPerBlock instances per workgroup, one per wavefront; they share nothing but the launch.
// Measured at 4 096 instances, N = 10: 231 / 230 / 225 / 242 us for 1 / 2 / 4 / 8 wavefronts per workgroup.
constexpr int kWavesPerBlock = 4;

// ---- the compact model's gradient by prefix sums (one horizon step per lane) --------------------------------
// The dense form above spends 2 (2H)^2 flops per iteration on a Hessian whose structure is a double integrator
// (A = [1 a; 0 1]): at N = 40 that is 12 800 flops for 1 824 algorithmic ones, 0.75 us per iteration, and a Hessian
// row in LDS that leaves a CU four wavefronts.  With lane i = horizon step i the two recurrences of mpc.h:275-281
// are SCANS over the lanes of affine maps with a constant matrix:
//   forward   [M0; M1][i] = A [M0; M1][i-1] + (a u1, c u0 - c u1)[i]          A^k   = [1 ka; 0 1]
//   backward  [N0; N1][i] = trans(A) [N0; N1][i+1] + (q0 M0, q1 M1)[i]        trans(A)^k = [1 0; ka 1]
// so "combine with the segment k lanes away" is one fma and two adds on values moved by DPP (row_shr / row_shl
// inside a 16-lane row, row_bcast:15 / :31 forward across rows, a lane read backward across rows): 6 steps per
// scan instead of 2H dependent ones, no Hessian, no LDS, ~40 registers -- the layout of wave2_solve (lane = step,
// both inputs in the lane) with its gradient replaced.  Every step is exact in the same sense as the sequential
// recurrence (sums of same-kind terms, no cancelling differences); the association differs, like everything in
// this family.  df = MM + trans(B) N + R u as in dlib (mpc.h:283), MM from the compact linear term.
// (From N = 40.  At N = 30 the dense one-variable-per-lane kernel is still ahead: 2.25 against 2.55 ms per 4 096
// instances, 6.9 against 7.5 per 16 384.  At N = 40: 4.7 / 4.7 / 15.0 ms per 2 048 / 4 096 / 16 384 instances against
// 7.8 / 13.0 / 42 with the Hessian rows -- 0.47 us per iteration instead of 0.75, and eight wavefronts per CU.)
template <typename T, int I, int H, class Model> constexpr bool wave_scan() {
    return I == 2 && std::is_same<Model, CompactModel<T>>::value && H >= 40 && H <= 48;
}
template <typename T> struct ScanConsts {   // per lane, set up once per instance
    T ka1, ka2, ka4, ka8;   // k a (wave-uniform)
    T m15, m31;             // (distance to lane 15 / 31 of the previous row / half) a
    T w1, d1, w0, d0;       // backward across rows: row 1 <- lane 32, row 0 <- lane 16 (weights 1/0, distance a)
    T q0a, q1a;             // Q on active lanes, 0 beyond the horizon
};
template <int H, typename T> TPC_DEV ScanConsts<T> scan_consts(T a, T q0, T q1, int lane) {
    ScanConsts<T> k;
    k.ka1 = a; k.ka2 = (T)2 * a; k.ka4 = (T)4 * a; k.ka8 = (T)8 * a;
    k.m15 = (T)((lane & 15) + 1) * a;
    k.m31 = (T)((lane & 31) + 1) * a;
    const bool r1 = lane >= 16 && lane < 32, r0 = lane < 16;
    k.w1 = r1 ? (T)1 : (T)0; k.d1 = r1 ? (T)(32 - lane) * a : (T)0;
    k.w0 = r0 ? (T)1 : (T)0; k.d0 = r0 ? (T)(16 - lane) * a : (T)0;
    k.q0a = lane < H ? q0 : (T)0; k.q1a = lane < H ? q1 : (T)0;
    return k;
}
// df[0], df[1] of this lane's horizon step from its controls u[0], u[1] (inactive lanes: u = 0)
template <int H, typename T> TPC_DEV void scan_gradient(const ScanConsts<T>& k, T a, T c, const T* r, const T* g,
                                                        const T* u, T* df) {
    // forward: inclusive prefix of the affine maps
    T pz = a * u[1];
    T py = tfma(c, u[0], -(c * u[1]));
    auto fwd = [&](auto ctrl, T ka) {
        constexpr int C = decltype(ctrl)::value;
        const T ys = dpp_shr0<C>(py), zs = dpp_shr0<C>(pz);
        pz = pz + tfma(ka, ys, zs);
        py = py + ys;
    };
    fwd(std::integral_constant<int, 0x111>{}, k.ka1);   // row_shr:1
    fwd(std::integral_constant<int, 0x112>{}, k.ka2);
    fwd(std::integral_constant<int, 0x114>{}, k.ka4);
    fwd(std::integral_constant<int, 0x118>{}, k.ka8);
    {   // rows 1, 3 <- lane 15 of the row before
        const T ys = dpp_mov<0x142, 0xa>((T)0, py), zs = dpp_mov<0x142, 0xa>((T)0, pz);
        pz = pz + tfma(k.m15, ys, zs);
        py = py + ys;
    }
    if constexpr (H > 32) {   // rows 2, 3 <- lane 31
        const T ys = dpp_mov<0x143, 0xc>((T)0, py), zs = dpp_mov<0x143, 0xc>((T)0, pz);
        pz = pz + tfma(k.m31, ys, zs);
        py = py + ys;
    }
    // backward: inclusive suffix of the affine maps over (q0 M0, q1 M1), zero beyond the horizon
    T n0 = k.q0a * pz, n1 = k.q1a * py;
    auto bwd = [&](auto ctrl, T ka) {
        constexpr int C = decltype(ctrl)::value;
        const T s0 = dpp_shr0<C>(n0), s1 = dpp_shr0<C>(n1);
        n1 = n1 + tfma(ka, s0, s1);
        n0 = n0 + s0;
    };
    bwd(std::integral_constant<int, 0x101>{}, k.ka1);   // row_shl:1
    bwd(std::integral_constant<int, 0x102>{}, k.ka2);
    bwd(std::integral_constant<int, 0x104>{}, k.ka4);
    bwd(std::integral_constant<int, 0x108>{}, k.ka8);
    if constexpr (H > 32) {   // row 1 <- the total of row 2 (its first lane)
        const T t0 = read_lane(n0, 32), t1 = read_lane(n1, 32);
        n1 = tfma(k.w1, t1, tfma(k.d1, t0, n1));
        n0 = tfma(k.w1, t0, n0);
    }
    {   // row 0 <- the total of everything behind it (first lane of row 1, just completed)
        const T t0 = read_lane(n0, 16), t1 = read_lane(n1, 16);
        n1 = tfma(k.w0, t1, tfma(k.d0, t0, n1));
        n0 = tfma(k.w0, t0, n0);
    }
    df[0] = tfma(c, n1, tfma(r[0], u[0], g[0]));                    // mpc.h:283: MM + trans(B) N + R u
    df[1] = tfma(a, n0, tfma(-c, n1, tfma(r[1], u[1], g[1])));
}

// ---- two decision variables per lane ------------------------------------------------------------------------
// Horizons whose I*H exceeds the wavefront (N = 40 with two inputs: 80 variables).  Lane i owns BOTH inputs of
// horizon step i: two Hessian rows (4H doubles: 320 registers at H = 40, one wavefront per SIMD), two controls,
// two of everything per-variable; the gradient is four v_fmac rows per source row (destination input x source
// input).  dlib scans the variables step by step, input 0 before input 1 (mpc.h:292-308), so the arg-max is the
// lane's own better variable (input 1 only if strictly larger) and then the same tagged-key maximum over lanes.
// Same arithmetic, verdicts and loop structure as wave_solve; 0.5-0.7 us per iteration (fp32 / fp64 at H = 40)
// against 5.1 us of a LANE lane.
template <typename T, int H> constexpr bool wave2_row_in_lds() { return 4 * H * (int)sizeof(T) > 960; }   // two rows past 240 registers
template <typename T, int H, class Model, class Args, class Hook = NoHook>
TPC_DEV void wave2_solve(const Args& g, const Knobs& kn, int64_t k, T* s_w, T* s_row1, Hook before_loops = Hook{}) {
    constexpr int I = 2, NL = H;
    // fp64 at H = 40: two rows are 320 registers, VALU operands must be architectural VGPRs (256) -- the second
    // row lives in LDS as [step][lane][2] (one ds_read_b128 per source step and iteration)
    constexpr bool SCAN = wave_scan<T, I, H, Model>();   // the gradient by prefix sums: no Hessian rows at all
    constexpr bool R1L = wave2_row_in_lds<T, H>() && !SCAN;
    static_assert(NL <= kWave && NL > 16, "one horizon step per lane, more than one row of lanes");
    const int lane = threadIdx.x & (kWave - 1);
    const bool active = lane < NL;
    const int qi = active ? lane : 0;

    Model m;
    m.load(g, k);
    const bool nonfinite = m.nonfinite();
    const bool badmodel = m.invalid();

    T row[(R1L || SCAN) ? 1 : 2][SCAN ? 1 : 2 * H];
    T my_qd[2] = {(T)0, (T)0}, my_g[2] = {(T)0, (T)0}, my_r[2];
    // the linear term first: its intermediates (identical in all lanes) are parked in LDS, and where the second
    // Hessian row goes to LDS they borrow its place before it is written (40 KB per wavefront is a quarter of the
    // CU's LDS exactly: a vector of its own beside it would cost the CU its fourth wavefront)
    {
        T* lt = R1L ? s_row1 : s_w;
        auto take_g = [&](int q, T val) { if (q == 2 * qi) my_g[0] = val; if (q == 2 * qi + 1) my_g[1] = val; };
        if constexpr (std::is_same<Model, CompactModel<T>>::value) linear_term<T, I, H>(m, (T*)nullptr, take_g);
        else linear_term_fn<T, I, H>(m, [&](int q, T val) { lt[q] = val; }, [&](int q) { return lt[q]; }, take_g);
        if (!active) my_g[0] = my_g[1] = (T)0;
    }
    T scan_lambda = (T)0;
    if constexpr (SCAN) {
        // no Hessian: Q_diag and lambda straight from dlib's constructor recurrence (mpc.h:116-123), every lane
        // keeping the pair of its own step
        scan_lambda = ctor_lambda_qdiag<T, I, H>(m, [&](int i, int j, T val) { if (i == qi) my_qd[j] = val; });
        if (!active) my_qd[0] = my_qd[1] = (T)0;
    } else if constexpr (R1L) {
        T tmp[2 * H];
        hessian_row<T, I, H>(m, active, qi, 1, tmp);
        my_qd[1] = active ? pick_own<T, 2 * H>(tmp, 2 * qi + 1) : (T)0;
#pragma unroll
        for (int l = 0; l < H; ++l) { s_row1[(l * kWave + lane) * 2] = tmp[2 * l]; s_row1[(l * kWave + lane) * 2 + 1] = tmp[2 * l + 1]; }
        hessian_row<T, I, H>(m, active, qi, 0, row[0]);
        my_qd[0] = active ? pick_own<T, 2 * H>(row[0], 2 * qi) : (T)0;
    } else {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            hessian_row<T, I, H>(m, active, qi, e, row[e]);
            my_qd[e] = active ? pick_own<T, 2 * H>(row[e], 2 * qi + e) : (T)0;
        }
    }
    my_r[0] = active ? m.R(0) : (T)0;
    my_r[1] = active ? m.R(1) : (T)0;
    // Everything below that is the same in all lanes -- one instance per wavefront: its bounds, and whatever follows
    // from the Hessian's trace -- is handed to the scalar registers (a register move, the value unchanged): the two
    // Hessian rows and the state leave the general model's kernel no vector register to spare, and what did not fit
    // was reloaded from scratch inside the iteration loops.
    T lambda_v;
    if constexpr (SCAN) lambda_v = scan_lambda;
    else lambda_v = wave_sum_all<NL>((my_qd[0] + my_r[0]) + (my_qd[1] + my_r[1]));
    const T lambda = wave_uniform(lambda_v);
    const T lo[2] = {wave_uniform(m.lo(0)), wave_uniform(m.lo(1))}, hi[2] = {wave_uniform(m.hi(0)), wave_uniform(m.hi(1))};
    const T eps = (T)kn.eps;
    const T my_rqd[2] = {(T)1 / my_qd[0], (T)1 / my_qd[1]};
    const T inv_lambda = wave_uniform((T)1.0 / lambda);
    const T sq = tsqrt(lambda);
    const T beta = wave_uniform((sq - (T)1) / (sq + (T)1));

    T u[2], v[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        u[e] = active ? WaveIO<T, I, H, Args>::init_u(g, k, qi, e) : (T)0;
        v[e] = active ? WaveIO<T, I, H, Args>::init_v(g, k, qi, e) : (T)0;
    }
    bool mask_ok = false;
    if constexpr (Model::kFastStop) {
        const T mm_max = wave_max<NL>(active ? tmax(tabs(my_g[0]), tabs(my_g[1])) : (T)0);
        const bool start_inside = u[0] >= lo[0] && u[0] <= hi[0] && u[1] >= lo[1] && u[1] <= hi[1];
        const bool term_nan = my_g[0] != my_g[0] || my_g[1] != my_g[1];
        mask_ok = m.fast_stop_ok(mm_max, eps, lambda, H) && __ballot(active && (!start_inside || term_nan)) == 0ull;
    }
    constexpr T kHuge = (T)(sizeof(T) == 8 ? 0x1p600 : 0x1p100);
    const T nlo_h[2] = {-(lo[0] * kHuge), -(lo[1] * kHuge)}, hi_h[2] = {hi[0] * kHuge, hi[1] * kHuge};
    T huge_r = kHuge;
    asm volatile("" : "+v"(huge_r));

    [[maybe_unused]] ScanConsts<T> sk;
    [[maybe_unused]] T scan_a = (T)0, scan_c = (T)0;
    if constexpr (SCAN) {
        scan_a = wave_uniform(m.a); scan_c = wave_uniform(m.c);
        sk = scan_consts<H, T>(scan_a, m.q0, m.q1, lane);
    }
    // df[e] = sum over source steps l and source inputs e2 of row[e][2 l + e2] * u[e2] of lane l, + R u + g
    auto gradient_of = [&](const T* uu, T* df) {
        if constexpr (SCAN) {
            scan_gradient<H, T>(sk, scan_a, scan_c, my_r, my_g, uu, df);
        } else {
        T x[2][4];
#pragma unroll
        for (int e2 = 0; e2 < 2; ++e2) {
            const Swapped<T> p = swap_rows16(uu[e2], uu[e2]);
            x[e2][0] = p.a; x[e2][1] = p.b; x[e2][2] = p.a; x[e2][3] = p.b;
            if constexpr (NL > 32) {
                const Swapped<T> ev = swap_halves(p.a, p.a);
                const Swapped<T> od = swap_halves(p.b, p.b);
                x[e2][0] = ev.a; x[e2][2] = ev.b; x[e2][1] = od.a; x[e2][3] = od.b;
            }
        }
        if constexpr (!R1L) {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                T a0 = my_g[e], a1 = my_r[e] * uu[e];
#pragma unroll
                for (int e2 = 0; e2 < 2; ++e2) {
                    static_for_w<(NL + 15) / 16>([&](auto rc) {
                        constexpr int r = decltype(rc)::value;
                        constexpr int cnt = NL - 16 * r < 16 ? NL - 16 * r : 16;
                        fmac_row<cnt, 2>(a0, a1, x[e2][r], row[e] + 32 * r + e2);
                    });
                }
                df[e] = a0 + a1;
            }
        } else {
            // eight source steps at a time: the LDS reads of the second row's entries go out first and the first
            // row's sixteen multiply-adds for the same steps run under their latency
            T a0 = my_g[0], a1 = my_r[0] * uu[0], b0 = my_g[1], b1 = my_r[1] * uu[1];
            static_for_w<(NL + 7) / 8>([&](auto cc) {
                constexpr int c8 = decltype(cc)::value;
                constexpr int cnt = NL - 8 * c8 < 8 ? NL - 8 * c8 : 8;
                constexpr int off = (8 * c8) % 16, r = (8 * c8) / 16;
                T tmp[16];
#pragma unroll
                for (int l = 0; l < cnt; ++l) {
                    tmp[2 * l] = s_row1[((8 * c8 + l) * kWave + lane) * 2];
                    tmp[2 * l + 1] = s_row1[((8 * c8 + l) * kWave + lane) * 2 + 1];
                }
                fmac_row8<cnt, off, 2>(a0, a1, x[0][r], row[0] + 16 * c8);
                fmac_row8<cnt, off, 2>(a0, a1, x[1][r], row[0] + 16 * c8 + 1);
                fmac_row8<cnt, off, 2>(b0, b1, x[0][r], tmp);
                fmac_row8<cnt, off, 2>(b0, b1, x[1][r], tmp + 1);
            });
            df[0] = a0 + a1;
            df[1] = b0 + b1;
        }
        }
    };

    // (blocks of 4, not 5, coordinate-descent steps: with 5 the work-queue kernel's fp64 build ran out of AGPRs and
    // kept half of one Hessian entry in scratch, reloaded inside the loops -- and that build returned wrong controls
    // for one instance in nine while the scratch-free builds of the same source agree with dlib to 1e-13.  The
    // cause was not found; tests/test_build_artifacts.py refuses a two-per-lane kernel that uses scratch.)
    constexpr int kUnrollCd = 4, kUnrollPg = 4;
    uint32_t iter = 0;
    bool capped = true;
    auto run = [&](auto mask_tag) {
        constexpr bool MASK = decltype(mask_tag)::value;
        auto masked = [&](int e, T uu, T df) -> T {
            if constexpr (MASK) {
                const T g_lo = tfma(uu, huge_r, nlo_h[e]), g_hi = tfma(uu, -huge_r, hi_h[e]);
                return tmax(tmin(df, g_lo), -g_hi);
            } else {
                const bool blocked = (uu <= lo[e] && df > (T)0) || (uu >= hi[e] && df < (T)0);   // mpc.h:298-299
                return (active && !blocked) ? df : (T)0;
            }
        };
        const bool qd_nz[2] = {my_qd[0] != (T)0, my_qd[1] != (T)0};
        const T my_rqd0[2] = {qd_nz[0] ? my_rqd[0] : (T)0, qd_nz[1] ? my_rqd[1] : (T)0};
        unsigned long long go_mask = ~0ull;
        bool go_lane = true, last_take = false;
        int cnt = 0;
        const T eps_up = round_up_key(eps);
        auto cd_step = [&]() {
            T df[2], c[2], nu[2];
            gradient_of(u, df);
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const T cs = masked(e, u[e], df[e]);
                c[e] = tabs(cs);
                if constexpr (!MASK && sizeof(T) == 8) c[e] = c[e] == (T)__builtin_inf() ? (T)1.7976931348623157e308 : c[e];
                if constexpr (MASK) nu[e] = tmax(tmin(tfma(-my_rqd0[e], df[e], u[e]), hi[e]), lo[e]);
                else {
                    nu[e] = put_in_range(lo[e], hi[e], -(df[e] - my_qd[e] * u[e]) * my_rqd[e]);
                    nu[e] = qd_nz[e] ? nu[e] : u[e];
                }
                asm volatile("" : "+v"(nu[e]));
            }
            const bool second = c[1] > c[0];              // input 1 only if strictly larger (NaN never wins)
            const T cbest = second ? c[1] : c[0];
            bool take_lane;
            if constexpr (sizeof(T) == 8) {
                const T key = pack_key_abs(cbest, 63 - lane);
                const T mk = wave_max<NL>(key);
                const T thr = raw_max(mk, eps_up);
                take_lane = key >= thr;
                go_lane = mk >= eps_up;                                          // mpc.h:310-311
            } else {
                const T mx = wave_max<NL>(cbest == cbest ? cbest : (T)0);
                go_lane = !(mx < eps);
                const unsigned long long hit = __ballot(cbest == mx);
                take_lane = go_lane && lane == __ffsll((long long)hit) - 1;
            }
            last_take = take_lane && (second ? qd_nz[1] : qd_nz[0]);
            u[0] = (take_lane && !second) ? nu[0] : u[0];
            u[1] = (take_lane && second) ? nu[1] : u[1];
            cnt += go_lane ? 1 : 0;
        };
        auto pg_step = [&]() {
            T df[2], v_new[2], u_new[2];
            gradient_of(u, df);
            bool above = false;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const T cs = masked(e, u[e], df[e]);
                above = above || tabs(cs) >= eps;
                v_new[e] = clamp3(tfma(-inv_lambda, df[e], u[e]), lo[e], hi[e]);
                u_new[e] = clamp3(tfma(beta, v_new[e] - v[e], v_new[e]), lo[e], hi[e]);
            }
            go_mask = __ballot(above);                                           // mpc.h:310-311: 0 = stop
            const unsigned long long go_all = any_to_all(go_mask);
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                v[e] = lane_select(go_all, v_new[e], v[e]);
                u[e] = lane_select(go_all, u_new[e], u[e]);
            }
            iter += go_mask != 0ull ? 1u : 0u;
        };
        const uint32_t cd_end = kn.smo_iters < kn.max_iter ? kn.smo_iters : kn.max_iter;
        const uint32_t cd_left = cd_end > iter ? cd_end - iter : 0u;
        for (uint32_t blk = cd_left / kUnrollCd; blk;) {
#pragma unroll
            for (int r = 0; r < kUnrollCd; ++r) cd_step();
            go_mask = __ballot(go_lane);
            blk = go_mask != 0ull ? blk - 1 : 0u;
        }
        for (uint32_t rest = go_mask != 0ull ? cd_left % kUnrollCd : 0u; rest; --rest) cd_step();
        go_mask = __ballot(go_lane);
        iter += (uint32_t)__builtin_amdgcn_readfirstlane(cnt);
        if (go_mask == 0ull) { capped = false; return; }
        if (iter == kn.smo_iters && __ballot(last_take) != 0ull) { v[0] = u[0]; v[1] = u[1]; }   // mpc.h:330-334
        const uint32_t pg_left = kn.max_iter > iter ? kn.max_iter - iter : 0u;
        for (uint32_t blk = pg_left / kUnrollPg; blk;) {
#pragma unroll
            for (int r = 0; r < kUnrollPg; ++r) pg_step();
            blk = go_mask != 0ull ? blk - 1 : 0u;
        }
        for (uint32_t rest = go_mask != 0ull ? pg_left % kUnrollPg : 0u; rest; --rest) pg_step();
        if (go_mask == 0ull) capped = false;
    };
    before_loops();
    if ((Model::kScreen && nonfinite) || badmodel) capped = false;
    else if (mask_ok) run(std::true_type{});
    else run(std::false_type{});
    uint32_t f = 0;
    if (nonfinite) f |= 0x1u;
    if (badmodel) f |= 0x4u;
    if (capped) f |= 0x2u;
    WaveIO<T, I, H, Args>::report(g, f, iter);
#pragma unroll
    for (int e = 0; e < 2; ++e) WaveIO<T, I, H, Args>::write(g, k, active, qi, e, u[e], v[e], iter);
    if (g.flags) raise_flags(g.flags, f);
}

// one instance by the calling wavefront, whichever layout its size needs
template <int I, int H> constexpr bool wave_two_per_lane() { return I * H > kWave; }
// LDS a wavefront needs beside s_w (T elements): the second Hessian row of the two-per-lane fp64 kernels
template <typename T, int I, int H, class Model = void> constexpr int wave_row_lds() {
    if constexpr (wave_scan<T, I, H, Model>()) return 1;   // (no Hessian at all)
    else if constexpr (I * H > kWave) return wave2_row_in_lds<T, H>() ? 2 * H * kWave : 1;
    else return 1;
}
// LDS for the linear term's intermediates (T elements): none of its own where the row's place is borrowed
template <typename T, int I, int H, class Model = void> constexpr int wave_lt_lds() { return wave_row_lds<T, I, H, Model>() > 1 ? 1 : 2 * H; }
// (one where a wavefront parks a Hessian row in LDS: 40 KB each, and a workgroup's static LDS ends at 64 KB)
template <typename T, int I, int H, class Model = void> constexpr int waves_per_block() { return wave_row_lds<T, I, H, Model>() > 1 ? 1 : kWavesPerBlock; }
// (`keep`: a resident wave's WaveKeep, used by the one-variable-per-lane layout only)
template <typename T, int I, int H, class Model, class Args, class Hook = NoHook, class Keep = NoKeep>
TPC_DEV void wave_solve_any(const Args& g, const Knobs& kn, int64_t k, T* s_w, T* s_row1, Hook before_loops = Hook{},
                            Keep* keep = nullptr) {
    if constexpr (wave_scan<T, I, H, Model>()) wave2_solve<T, H, Model, Args, Hook>(g, kn, k, s_w, s_row1, before_loops);
    else if constexpr (I * H <= kWave) wave_solve<T, I, H, Model, Args, Hook, Keep>(g, kn, k, s_w, before_loops, keep);
    else wave2_solve<T, H, Model, Args, Hook>(g, kn, k, s_w, s_row1, before_loops);
}

// Registers: two wavefronts per SIMD is what the work queue keeps resident and what a batch that fits the
// chip at once amounts to, so that is the occupancy asked for (256 registers): only the 60-variable kernels'
// set-up code spills under it (a few dozen scratch accesses per instance, none inside a loop).  Asking for
// more made the set-up code of the smaller kernels spill too: 20 MB of scratch traffic per 4 096 instances.
constexpr int kWaveMinWaves = 2;
// (the general model's 60-variable fp64 kernels -- a Hessian row of 120 registers beside the per-instance model --
// do not fit 256 registers without spilling inside their loops: they get the whole file, one wavefront per SIMD)
template <typename T, int I, int H, class Model = void> constexpr int wave_min_waves() {
    constexpr bool big_general = sizeof(T) == 8 && I * H > 48 && !std::is_same<Model, CompactModel<T>>::value &&
                                 !std::is_same<Model, void>::value;
    if (wave_scan<T, I, H, Model>()) return kWaveMinWaves;   // ~40 registers: no reason for less
    return (wave_two_per_lane<I, H>() || big_general) ? 1 : kWaveMinWaves;
}
template <typename T, int I, int H, class Model, class Args>
__global__ __launch_bounds__((waves_per_block<T, I, H, Model>() * kWave)) __attribute__((amdgpu_waves_per_eu(wave_min_waves<T, I, H, Model>())))
void wave_kernel(Args g, Knobs kn) {
    constexpr int WPB = waves_per_block<T, I, H, Model>();
    __shared__ __attribute__((aligned(16))) T s_w[WPB][wave_lt_lds<T, I, H, Model>()];
    __shared__ __attribute__((aligned(16))) T s_row1[WPB][wave_row_lds<T, I, H, Model>()];
    const int w = threadIdx.x / kWave;
    const int64_t k = (int64_t)blockIdx.x * WPB + w;
    if (k < g.n) wave_solve_any<T, I, H, Model, Args>(g, kn, k, s_w[w], s_row1[w]);
}


// ---- work queue (batches with more instances than the persistent grid holds) -----------------------------
// Iteration counts differ ~30x between instances, a SIMD runs the wavefronts it was given and nothing else,
// and with every instance resident at once a launch lasts as long as its unluckiest SIMD: 1.8x the mean at
// 4 096 instances.  From N = 10 up the grid is therefore kQueueWorkgroupsPerCu workgroups of four persistent
// wavefronts per CU (two per SIMD), each taking instances from an atomic ticket over a queue ordered
// longest-first -- by lambda, dlib's own Hessian trace bound (the step is 1/lambda: large lambda, many small
// steps; Spearman 0.89 / 0.97 with the iteration count at N = 10 / 20), or by the caller's work hint.
// Measured (kernel time, 4 096 / 16 384 instances): N = 10: 226 -> 215 / 609 -> 602 us, N = 20: 1 266 -> 909 /
// 3 455 -> 2 575 us, N = 30: 3 849 -> 2 488 / 11 256 -> 7 164 us.  At N = 4 and 5 an instance is too short for the
// queue to pay (45 -> 92 us): those keep one launch slot per instance.
constexpr int kQueueWorkgroupsPerCu = 2;   // (one where a lane holds two variables: 380 registers per lane)
template <int I, int H, typename T = void, class Model = void> constexpr int queue_waves_per_cu() {
    return ((wave_two_per_lane<I, H>() && !wave_scan<T, I, H, Model>()) ? 1 : kQueueWorkgroupsPerCu) * kWavesPerBlock;
}
constexpr int kQueueMinHorizon = 10;
constexpr int kOrderThreads = 1024, kOrderBins = 2048, kOrderPerThread = 32;
constexpr int64_t kQueueMaxInstances = (int64_t)kOrderThreads * kOrderPerThread;   // larger batches: plain launch
static_assert(kQueueMaxInstances == kWaveQueueMaxInstances, "tpc_mpc_api.cpp's AUTO rule counts on it");

// One workgroup: a key per instance (kept in registers) -> counting sort over kOrderBins linear bins of the
// key range -> order[] (descending); the queue's ticket counters are zeroed here.  Positions inside a bin
// come from atomics and are not reproducible; they decide which wavefront solves which instance, never a result.
template <typename T, int I, int H, class Model, class Args>
__global__ __launch_bounds__(kOrderThreads) void wave_order_kernel(Args g, uint32_t* __restrict__ order,
                                                                   uint32_t* __restrict__ tickets) {
    __shared__ uint32_t bins[kOrderBins];
    __shared__ uint32_t wave_sum[kOrderThreads / kWave];
    __shared__ uint32_t kmin, kmax;
    const int t = threadIdx.x;
    const int n = (int)g.n;
    if (t == 0) { kmin = 0xffffffffu; kmax = 0u; }
    if (t < kQueueTickets) tickets[t * kQueueTicketStride] = 0u;
    for (int b = t; b < kOrderBins; b += kOrderThreads) bins[b] = 0u;
    uint32_t key[kOrderPerThread];
    uint32_t lo = 0xffffffffu, hi = 0u;
#pragma unroll
    for (int j = 0; j < kOrderPerThread; ++j) {
        const int i = j * kOrderThreads + t;
        key[j] = 0u;
        if (j * kOrderThreads < n) {
            if (i < n) {
                if (g.work_hint) {
                    const int32_t w = g.work_hint[i];
                    key[j] = w > 0 ? (uint32_t)w : 0u;
                } else {
                    Model m;
                    m.load(g, i);
                    float lf;
                    if constexpr (std::is_same<Model, CompactModel<T>>::value)
                        lf = (float)tabs(m.a);   // one batch-wide Q, R, T, l: lambda grows with |T v|, which orders the same
                    else {
                        // a sort key only: the trace bound in fp32 on an fp32 copy of the model (half the registers of
                        // the fp64 recurrence, which spilled inside this loop)
                        GeneralModel<float, I> mf;
                        mf.a00 = (float)m.a00; mf.a01 = (float)m.a01; mf.a10 = (float)m.a10; mf.a11 = (float)m.a11;
#pragma unroll
                        for (int jj = 0; jj < I; ++jj) { mf.b[0][jj] = (float)m.b[0][jj]; mf.b[1][jj] = (float)m.b[1][jj]; mf.r[jj] = (float)m.r[jj]; }
                        mf.q0 = (float)m.q0; mf.q1 = (float)m.q1;
                        lf = ctor_lambda_qdiag<float, I, H>(mf, [](int, int, float) {});
                    }
                    key[j] = (lf > 0.0f && lf < __builtin_inff()) ? (uint32_t)__float_as_int(lf) : 0u;
                }
                lo = key[j] < lo ? key[j] : lo;
                hi = key[j] > hi ? key[j] : hi;
            }
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        const uint32_t l2 = (uint32_t)__shfl_xor((int)lo, off), h2 = (uint32_t)__shfl_xor((int)hi, off);
        lo = l2 < lo ? l2 : lo;
        hi = h2 > hi ? h2 : hi;
    }
    __syncthreads();
    if ((t & 63) == 0) { atomicMin(&kmin, lo); atomicMax(&kmax, hi); }
    __syncthreads();
    const uint32_t top = kmax, span = kmax > kmin ? kmax - kmin : 1u;
    auto bin_of = [&](uint32_t k) { return (uint32_t)(((uint64_t)(top - k) * (kOrderBins - 1)) / span); };
#pragma unroll
    for (int j = 0; j < kOrderPerThread; ++j)
        if (j * kOrderThreads + t < n) atomicAdd(&bins[bin_of(key[j])], 1u);
    __syncthreads();
    constexpr int kPer = kOrderBins / kOrderThreads;
    uint32_t c[kPer], sum = 0;
    for (int j = 0; j < kPer; ++j) { c[j] = bins[t * kPer + j]; sum += c[j]; }
    uint32_t incl = sum;
    for (int off = 1; off < kWave; off <<= 1) {
        const uint32_t up = (uint32_t)__shfl_up((int)incl, off);
        if ((t & 63) >= off) incl += up;
    }
    if ((t & 63) == 63) wave_sum[t >> 6] = incl;
    __syncthreads();
    uint32_t pos = incl - sum;
    for (int w = 0; w < (t >> 6); ++w) pos += wave_sum[w];
    for (int j = 0; j < kPer; ++j) { bins[t * kPer + j] = pos; pos += c[j]; }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < kOrderPerThread; ++j)
        if (j * kOrderThreads + t < n) order[atomicAdd(&bins[bin_of(key[j])], 1u)] = (uint32_t)(j * kOrderThreads + t);
}

// The same queue walked in groups of G (2 or 4): position p stands for the instances order[G p] .. order[G p + G - 1]
// -- neighbours of the longest-first order, i.e. alike in length -- solved by one wavefront (wave_pair_solve).
template <typename T, int I, int H, int G, class Model, class Args>
__global__ __launch_bounds__((kWavesPerBlock * kWave)) __attribute__((amdgpu_waves_per_eu(2)))
void wave_pair_queue_kernel(Args g, Knobs kn, const uint32_t* __restrict__ order, uint32_t* tickets) {
    __shared__ __attribute__((aligned(16))) T s_w[kWavesPerBlock][G * 2 * H];
    const int w = threadIdx.x / kWave;
    const uint32_t n = (uint32_t)g.n, groups = (n + (uint32_t)G - 1u) / (uint32_t)G;
    const uint32_t waves = gridDim.x * kWavesPerBlock, wid = blockIdx.x * kWavesPerBlock + w;
    const uint32_t sub = wid % kQueueTickets;
    uint32_t* my_ticket = tickets + sub * kQueueTicketStride;
    uint32_t t = wid;
    bool first = true;
    while (t < groups) {
        const bool dynamic = !first && 2u * waves < groups;
        uint32_t drawn = 0;
        auto ask = [&]() {
            if (dynamic && (threadIdx.x & (kWave - 1)) == 0) drawn = atomicAdd(my_ticket, 1u);
        };
        int64_t ks[G];
#pragma unroll
        for (int j = 0; j < G; ++j) {
            const uint32_t pos = (uint32_t)G * t + (uint32_t)j;
            ks[j] = pos < n ? (int64_t)order[pos] : (int64_t)-1;
        }
        wave_pair_solve<T, I, H, G, Model, Args>(g, kn, ks, s_w[w], ask);
        const uint32_t d = (uint32_t)__builtin_amdgcn_readfirstlane((int)drawn);
        t = first ? 2u * waves - 1u - wid : (dynamic ? 2u * waves + d * kQueueTickets + sub : groups);
        first = false;
    }
}

// W persistent wavefronts over the queue.  The first two rounds are static: wavefront w takes position w, then
// position 2W-1-w -- the longest instance is paired with the shortest of the second round, and so on inwards.
// What lies beyond 2W is dealt out dynamically, through kQueueTickets counters instead of one: returning atomics
// on ONE address complete at about one per 30-38 ns on this chip whatever the number of waves asking (measured:
// 32 768 empty solves with one ticket each took 1.1 ms, and so did 32 768 real N = 10 solves), which capped the
// queue at ~30 M instances/s and opened every launch with a 2W-deep burst.  Counter j (wavefronts w = j mod
// kQueueTickets) hands out positions 2W + j, 2W + j + kQueueTickets, ...: the sub-queues interleave, so each
// is longest-first with the same mix and they run dry within an instance or two of each other.  A wavefront
// asks for its next position when the set-up of the current instance is done, just before its iteration loops
// (see wave_solve: any earlier and the in-order wait counter makes the set-up's own loads wait for the ticket).
template <typename T, int I, int H, class Model, class Args>
__global__ __launch_bounds__((waves_per_block<T, I, H, Model>() * kWave)) __attribute__((amdgpu_waves_per_eu(wave_min_waves<T, I, H, Model>())))
void wave_queue_kernel(Args g, Knobs kn, const uint32_t* __restrict__ order, uint32_t* tickets) {
    constexpr int WPB = waves_per_block<T, I, H, Model>();
    __shared__ __attribute__((aligned(16))) T s_w[WPB][wave_lt_lds<T, I, H, Model>()];
    __shared__ __attribute__((aligned(16))) T s_row1[WPB][wave_row_lds<T, I, H, Model>()];
    const int w = threadIdx.x / kWave;
    const uint32_t n = (uint32_t)g.n;
    const uint32_t waves = gridDim.x * WPB, wid = blockIdx.x * WPB + w;
    const uint32_t sub = wid % kQueueTickets;
    uint32_t* my_ticket = tickets + sub * kQueueTicketStride;
    uint32_t t = wid;
    bool first = true;
    while (t < n) {
        const bool dynamic = !first && 2u * waves < n;
        uint32_t drawn = 0;   // (turned into a position only after the solve: its first use is where the wait goes)
        auto ask = [&]() {
            if (dynamic && (threadIdx.x & (kWave - 1)) == 0) drawn = atomicAdd(my_ticket, 1u);
        };
        wave_solve_any<T, I, H, Model, Args>(g, kn, (int64_t)order[t], s_w[w], s_row1[w], ask);
        const uint32_t d = (uint32_t)__builtin_amdgcn_readfirstlane((int)drawn);
        t = first ? 2u * waves - 1u - wid : (dynamic ? 2u * waves + d * kQueueTickets + sub : n);
        first = false;
    }
}

}  // namespace tpc
