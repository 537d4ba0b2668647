// LANE_FMA kernels: one lane per MPC instance like the LANE family (mpc_lane.h), in the unit-box /
// fused-multiply-add arithmetic of mpc_ub_model.h -- the tolerance-grade throughput family.
//
// Same three launches per batch and the same scratch as LANE (records, queue keys, counting sort,
// ticket): ub_cd_kernel (coordinate descent, mpc.h:319-335), the queue order (mpc_sort.hip),
// ub_pg_kernel (accelerated projected gradient with lane refill, mpc.h:336-345).  What differs is
// the arithmetic: an iteration of the projected-gradient kernel is ~25 VALU instructions per
// horizon step instead of 49 (see mpc_ub_model.h for why), nothing of the model lives in memory
// any more (no linear term MM), and the only LDS traffic left is dlib's momentum vector v.
// Compact model only (the general form keeps the LANE family).
//
// Results: identical decisions on quantities that differ from dlib's by rounding: max |du| vs dlib
// ~2e-13 at N = 20 (1e-12 at N = 40), identical iteration counts on every instance of the BASELINE
// workloads; bit-identical to the CPU model tests/model/ub_model.cpp.
#pragma once

#include "mpc_lane.h"
#include "mpc_ub_model.h"

namespace tpc {

template <typename T, bool EQB> TPC_DEV void ub_set_uniform(ub::Unit<T, EQB>& m, T gscale, const CompactArgs& g) {
    const T q[2] = {(T)g.q[0], (T)g.q[1]}, r[2] = {(T)g.r[0], (T)g.r[1]};
    const T lo[2] = {(T)g.lo[0], (T)g.lo[1]}, hi[2] = {(T)g.hi[0], (T)g.hi[1]};
    m.set_uniform(gscale, q, r, lo, hi);
}

// ------------------------------------------------------------------------------------------------
// Phase 1: coordinate descent, 64 instances per wave in lockstep (see lane_cd_kernel).  In LDS ([var][lane]): the
// reciprocals 1 / (Q_diag s), read at the arg-max's index, and -- where it fits a workgroup's 64 KB -- x itself (UbCdPlan).
//
// One iteration is a forward pass and a backward sweep that forms df[i] and offers it to the running arg-max at once
// (no gradient is stored).  dlib scans i then j ascending with a strict '>' (mpc.h:289-309): the lowest index among
// equal maxima wins; the sweep runs i and j DESCENDING with '>=', which picks the same one.  Two builds of the sweep:
//   exact   dlib's mask by compare and select (mpc.h:298-299), the running maximum, its index, sign and x: 15
//           instructions per variable, half of them halves of 64-bit selects;
//   fast    (every lane of the wavefront passed the screen of the select-free stop test, ub::fast_stop_ok) the mask as
//           arithmetic: g_lo / g_hi are 0 exactly on the bound and beyond every |df| off it (unit box: ldexp(x, 1900)
//           and ldexp(1 - x, 1900), which cannot underflow to 0 for any x > 0; dlib's coordinates: the gap times 2^100
//           as in ub_pg_kernel), mm = max(min(df, g_lo), -g_hi) is df where dlib counts it and 0 where it does not, and
//           ONE signed value tracks the arg-max: better = |mm| >= |best|, best = better ? mm : best.  Same decisions,
//           same x: both builds are bit-identical wherever the screen holds (the GPU tests hold them to the CPU model,
//           which implements dlib's mask).
// The winner's x comes from the LDS copy by index, and its update goes there; where registers are short (fp64, N >= 20)
// the passes read x from LDS too and no register copy exists -- the 120-instruction select chain that wrote one
// element of a register array, and ~240 AGPR moves per iteration, are gone.
template <typename T, int H> struct UbCdPlan {
    // x[2H][64] beside s_iqd[2H][64]: within a workgroup's 64 KB of static LDS, and without costing residency (the
    // 4 x CdOcc wavefronts of a CU share 160 KB: fp64 N = 30 would keep two of four SIMDs idle)
    static constexpr bool mirror = 2 * (2 * H * kWave * (int)sizeof(T)) <= 64 * 1024 &&
                                   2 * (2 * H * kWave * (int)sizeof(T)) * 4 * CdOcc<T, H>::value <= 160 * 1024;
    static constexpr bool x_in_lds = mirror && sizeof(T) == 8 && H >= 20;
    static constexpr bool fast = true;   // (the select-free sweep wherever the screen allows it)
};

TPC_DEV double cd_ldexp(double x, int e) { return __builtin_ldexp(x, e); }
TPC_DEV float cd_ldexp(float x, int e) { return __builtin_ldexpf(x, e); }

template <typename T, int H, bool EQB>
__global__ __launch_bounds__(64, (CdOcc<T, H>::value)) void ub_cd_kernel(CompactArgs g, Knobs kn, T* __restrict__ recs,
                                                                         uint32_t* __restrict__ keys,
                                                                         uint32_t* __restrict__ key_rank,
                                                                         uint32_t* __restrict__ key_hist,
                                                                         unsigned long long* __restrict__ stats) {
    using P = UbCdPlan<T, H>;
    constexpr int RL = LaneRec<T, H>::kLen;
    constexpr bool MIRROR = P::mirror, XL = P::x_in_lds;
    __shared__ T s_iqd[2 * H][kWave];
    __shared__ T s_x[MIRROR ? 2 * H : 1][kWave];
    const int lane = threadIdx.x;
    const int64_t k = (int64_t)blockIdx.x * kWave + lane;
    if (k >= g.n) return;

    ub::Unit<T, EQB> m;
    ub_set_uniform(m, (T)1, g);
    const T ty = ((const T*)g.dy)[k], tphi = ((const T*)g.dphi)[k];
    m.set_instance((T)g.step, (T)g.wheelbase, ((const T*)g.v)[k], ty, tphi);
    const bool nonfinite = m.nonfinite_inputs(ty, tphi);
    const T q0 = (T)g.q[0], q1 = (T)g.q[1], r0 = (T)g.r[0], r1 = (T)g.r[1];

    T x[XL ? 1 : 2 * H], w[ub::Reverse<T, H>::value ? 1 : 2 * H];
    auto X = [&](int q) -> T { if constexpr (XL) return s_x[q][lane]; else return x[q]; };
#pragma unroll
    for (int q = 0; q < 2 * H; ++q) {
        if constexpr (!XL) x[q] = m.xz(q & 1);
        if constexpr (MIRROR) s_x[q][lane] = m.xz(q & 1);
    }
    const T lambda = ub::ctor_lambda_qdiag<T, H>(m.a, m.c, q0, q1, r0, r1, [&](int i, int j, T val) {
        s_iqd[2 * i + j][lane] = val != (T)0 ? (T)1 / (val * m.s(j)) : (T)0;   // mpc.h:322: a zero Q_diag never updates
    });
    const T eps = (T)kn.eps;
    const bool screen_ok = ub::fast_stop_ok(m, ty, tphi, q0, q1, r0, r1, eps, lambda);
    const unsigned long long failing = __ballot(!screen_ok);
    if (failing != 0ull && lane == __ffsll((long long)failing) - 1) atomicOr(&stats[2], 1ull);
    if constexpr (sizeof(T) == 4) {   // fp32: may the projected-gradient kernel read its stop test off the projected step?
        const unsigned long long unmoved = __ballot(!ub::moved_stop_ok(m, eps, lambda));
        if (unmoved != 0ull && lane == __ffsll((long long)unmoved) - 1) atomicOr(&stats[2], 2ull);
    }
    const uint32_t cd_iters = kn.smo_iters < kn.max_iter ? kn.smo_iters : kn.max_iter;
    uint32_t iter = 0;
    bool stopped = nonfinite;
    bool vinit = false;
    constexpr T kHuge = (T)(sizeof(T) == 8 ? 0x1p600 : 0x1p100);
    T huge = kHuge;
    asm volatile("" : "+v"(huge));

    auto iteration = [&](auto fast_tag, uint32_t it) {
        constexpr bool F = decltype(fast_tag)::value;
        constexpr bool RV = ub::Reverse<T, H>::value;
        T Z, Y;
        m.fwd_init(Z, Y);
#pragma unroll
        for (int i = 0; i < H; ++i) {
            m.fwd(Z, Y, X(2 * i), X(2 * i + 1));
            if constexpr (!RV) { w[2 * i] = Z; w[2 * i + 1] = Y; }
        }
        T n0, n1;
        m.bwd_last(n0, n1, Z, Y);
        T max_df = (T)0, best_x = (T)0, best_mm = (T)0;
        int best = 0, best_sign = 0;
#pragma unroll
        for (int i = H - 1; i >= 0; --i) {
            if (i < H - 1) {
                if constexpr (RV) m.bwd(n0, n1, Z, Y); else m.bwd(n0, n1, w[2 * i], w[2 * i + 1]);
            }
            const T xi[2] = {X(2 * i), X(2 * i + 1)};
#pragma unroll
            for (int j = 1; j >= 0; --j) {
                const int q = 2 * i + j;
                const T xx = xi[j];
                const T dd = j == 0 ? m.df0(n1, xx) : m.df1(n0, n1, xx);
                if constexpr (F) {
                    T g_lo, g_hi;
                    if constexpr (ub::Unit<T, EQB>::kUnitBox) { g_lo = cd_ldexp(xx, 1900); g_hi = cd_ldexp((T)1 - xx, 1900); }
                    else { g_lo = m.gap_lo(j, xx, huge); g_hi = m.gap_hi(j, xx, huge); }
                    const T mm = tmax(tmin(dd, g_lo), -g_hi);         // df where dlib counts it (mpc.h:298-299), else 0
                    const bool better = tabs(mm) >= tabs(best_mm);   // (zeros may pass one another: all of them mean 'none')
                    best_mm = better ? mm : best_mm;
                    best = better ? q : best;
                    if constexpr (!MIRROR) best_x = better ? xx : best_x;
                } else {
                    const T up = (xx <= m.bl(j)) ? (T)0 : dd;
                    const T dn = (xx >= m.bh(j)) ? (T)0 : -dd;
                    const T mag = tmax(up, dn);
                    const bool better = mag >= max_df && mag > (T)0;   // (a zero never displaces the initial "none")
                    max_df = tmax(max_df, mag);
                    best = better ? q : best;
                    best_sign = better ? sign_word(dd) : best_sign;
                    if constexpr (!MIRROR) best_x = better ? xx : best_x;
                }
            }
            if constexpr (RV) { if (i > 0) m.rev(Z, Y, xi[0], xi[1]); }
        }
        if constexpr (F) max_df = tabs(best_mm);
        // (select form, no divergent block: a conditional update of the register copy of x costs a copy of all of it)
        stopped = stopped || max_df < eps;                      // mpc.h:310-311
        const bool act = !stopped;
        T best_df;
        if constexpr (F) best_df = best_mm; else best_df = with_sign(max_df, best_sign);
        if constexpr (MIRROR) best_x = s_x[best][lane];
        const T iq = s_iqd[best][lane];
        const bool upd = act && iq != (T)0;                     // mpc.h:322 (`continue` still counts)
        const T nx = m.project(ub::fma_(-iq, best_df, best_x), best & 1);   // mpc.h:325-326
        if constexpr (MIRROR) { if (upd) s_x[best][lane] = nx; }
        if constexpr (!XL) {
            const int sel = upd ? best : -1;
#pragma unroll
            for (int q = 0; q < 2 * H; ++q) x[q] = (q == sel) ? nx : x[q];
        }
        vinit = upd ? (it + 1 == kn.smo_iters) : vinit;         // mpc.h:330-334
        iter += act ? 1u : 0u;
    };
    if (P::fast && failing == 0ull) {
#pragma unroll 1
        for (uint32_t it = 0; it < cd_iters; ++it) {
            if (__ballot(!stopped) == 0ull) break;
            iteration(std::true_type{}, it);
        }
    } else {
#pragma unroll 1
        for (uint32_t it = 0; it < cd_iters; ++it) {
            if (__ballot(!stopped) == 0ull) break;
            iteration(std::false_type{}, it);
        }
    }

    T* rec = recs + (int64_t)k * RL;
#pragma unroll
    for (int q = 0; q < 2 * H; ++q) rec[q] = X(q);
    rec[2 * H] = lambda;
    uint64_t meta = (uint64_t)iter;
    if (stopped) meta |= kMetaStopped;
    if (vinit) meta |= kMetaVInit;
    if (nonfinite) meta |= kMetaNonFinite;
    store_meta<T>(rec + 2 * H + 1, meta);
    {   // what a refill pass of the projected-gradient kernel would otherwise compute per instance with a handful of
        // its lanes -- two divisions and a square root for the step constants (mpc.h:342-343), one for c -- and fetch
        // from three more arrays: computed here by all 64 lanes at once and left in the record (same operations, same bits)
        T il0, il1, beta;
        ub::pg_constants<T>(lambda, m.s0, m.s1, il0, il1, beta);
        T* ex = rec + LaneRec<T, H>::kExtra;
        ex[0] = il0; ex[1] = il1; ex[2] = beta; ex[3] = m.a; ex[4] = m.c; ex[5] = ty; ex[6] = tphi;
    }
    // queue key: as lane_cd_kernel (longest first by lambda, the floor rule, finished instances last)
    const bool finished = stopped || iter >= kn.max_iter;
    const T lambda_floor = (r0 + r1) * (T)H;
    const bool uninformative = lambda < (T)1.5 * lambda_floor;
    const float lf = g.work_hint ? (float)(g.work_hint[k] > 0 ? g.work_hint[k] : 1) : (float)lambda;
    const uint32_t spread = (uint32_t)k & 127u;
    uint32_t key = __float_as_uint(lf);
    if (finished) key = spread << 16;
    else if (!g.work_hint && uninformative) key = (0x7f00u + spread) << 16;
    else if (!(lf > 0.0f) || key < 0x00800000u) key = 0x00800000u;
    else if (key >= 0x7f000000u) key = 0x7effffffu;
    uint32_t f = 0;
    if (finished) {   // complete: published here, the PG kernel's queue ends before these
        ((T*)g.front)[k] = nonfinite ? (T)0 : m.control(0, X(0));
        ((T*)g.rear)[k] = nonfinite ? (T)0 : m.control(1, X(1));
        if (g.iters) g.iters[k] = (int32_t)iter;
        if (nonfinite) f |= 0x1u;
        if (!stopped) f |= 0x2u;
    }
    raise_flags(g.flags, f);
    keys[k] = key;
    key_rank[k] = atomicAdd(&key_hist[key >> 16], 1u);
}

// ------------------------------------------------------------------------------------------------
// Phase 2: the fused projected-gradient kernel.  Structure and refill protocol are those of
// lane_pg_fused_kernel (mpc_lane.h), where every design decision is explained; here only what differs.
//
// State of one instance: x (2H), the forward pass (Z, Y per step: 2H), dlib's momentum v (2H).
// Where it lives (UbPlan): x always in VGPRs; the forward pass in VGPRs, or -- where x and it do
// not both fit the 256 a VALU instruction can name (fp64, N >= 30) -- not kept at all: the backward sweep
// regenerates each step's (Z, Y) from the next one's (ub::Reverse); v in VGPRs for the first KV steps, in LDS beyond
// ([var][lane] columns, fetched one step ahead).
#ifndef TPC_UB_OCC
#define TPC_UB_OCC 0
#endif
#ifndef TPC_UB_KV
#define TPC_UB_KV -1
#endif
template <typename T, int H> struct UbPlan {
    static constexpr bool D = sizeof(T) == 8;
    static constexpr int words = D ? 2 : 1;
    // everything in registers: 6H values
    static constexpr bool reverse = ub::Reverse<T, H>::value;   // no stored forward pass (mpc_ub_model.h)
    static constexpr bool regs = 6 * H * words <= 200 || !D || (reverse && 4 * H * words <= 200);
    // waves per SIMD.  fp64: ONE at every horizon -- two were measured slower wherever they fit (N = 4 / 5 / 10:
    // 0.237 / 0.260 / 0.963 ms per 262 144 instances against 0.195 / 0.236 / 0.929 with one; N = 20 with v in
    // LDS or with the forward pass regenerated: 6.9-7.7 ms against 5.35): a second wave adds no issue slots
    // to a stream of 3-operand fp64 instructions, and halves the instances per lane of the persistent grid.
    // fp32 (measured per horizon, 262 144 instances, one / two waves per SIMD): N = 4: 0.194 / 0.257 ms, N = 5:
    // 0.241 / 0.260, N = 10: 0.943 / 0.788, N = 20: 5.58 / 5.26, N = 30: 18.4 / 19.5 -- two only at N = 10 and 20.
    static constexpr int occ_default = D ? 1 : ((H == 10 || H == 20) ? 2 : 1);
    static constexpr int occ = TPC_UB_OCC > 0 ? TPC_UB_OCC : occ_default;
    // steps of v in VGPRs (the rest in LDS)
    // (N = 30 / 40 with the forward pass regenerated, kernel time per 262 144 instances at KV = 0 / 8 / 16:
    // 24.3 / 22.7 / 22.9 ms and 58.9 / 55.5 / 54.0 ms on the box where the checkpointed plan took 23.7 and 55.9)
    static constexpr int kv_default = regs ? H : (H == 20 ? 16 : (H == 30 ? 8 : 16));
    static constexpr int kv = regs ? H : (TPC_UB_KV >= 0 ? TPC_UB_KV : kv_default);
    static constexpr bool step_barrier = H - kv >= 8;
};
template <int H> struct UbRefillBatch { static constexpr int value = RefillBatch<H>::value; };

// MODE: which stop test (the coordinate-descent kernel's screens pick one build per batch, stats[2]):
//   0 exact  dlib's mask by compare and select (mpc.h:298-299); any input
//   1 mask   fp32 only: dlib's mask as arithmetic (gaps times 2^100, v_med3_f32)
//   2 moved  min(|g df|, |x - x_new|) on the scaled gradient (fp64: the unit-box form; fp32: where ub::moved_stop_ok holds)
template <typename T, int H, bool EQB, int MODE>
__global__ __launch_bounds__((64 * UbPlan<T, H>::occ), (UbPlan<T, H>::occ)) void ub_pg_kernel(
    CompactArgs g, Knobs kn, const T* __restrict__ recs, const uint32_t* __restrict__ order,
    uint32_t* __restrict__ ticket, unsigned long long* __restrict__ stats, const uint32_t* __restrict__ queue_len) {
    using P = UbPlan<T, H>;
    constexpr int RL = LaneRec<T, H>::kLen;
    const int64_t n_queue = (int64_t)__builtin_nontemporal_load(queue_len);
    constexpr bool FAST = MODE != 0;
    static_assert(MODE == 0 || MODE == 2 || (MODE == 1 && sizeof(T) == 4), "the mask-as-arithmetic build is fp32's");
    {   // the builds are launched back to back; the coordinate-descent kernel's screens picked one
        const unsigned long long sel = __builtin_nontemporal_load(&stats[2]);
        const int need = (sel & 1ull) ? 0 : ((sizeof(T) == 4 && (sel & 2ull)) ? 1 : 2);
        if (need != MODE) return;
        if (n_queue <= 0) return;   // (nothing queued: no wavefront should go and ask the ticket -- a thousand returning atomics on one address take 35 us)
    }
    constexpr int BT = kWave * P::occ;
    constexpr int KV = P::kv, VL = H - KV;
    __shared__ T s_all[VL > 0 ? 2 * VL : 1][BT];
    T r_v[2 * KV + 1];
    const int lane = threadIdx.x;
    auto v_put = [&](int q, T val) { if (q < 2 * KV) r_v[q] = val; else s_all[q - 2 * KV][threadIdx.x] = val; };
    auto v_get = [&](int q) -> T { if (q < 2 * KV) return r_v[q]; else return s_all[q - 2 * KV][threadIdx.x]; };
    constexpr bool RV = P::reverse;
    T w[RV ? 1 : 2 * H];
    auto w_put = [&](int i, T Z, T Y) {
        if constexpr (!RV) { w[2 * i] = Z; w[2 * i + 1] = Y; }
    };
    auto w_getz = [&](int i) -> T { return w[2 * i]; };
    auto w_gety = [&](int i) -> T { return w[2 * i + 1]; };

    constexpr T gs = ub::GradScale<T>::g;
    const T geps = gs * (T)kn.eps;
    // MODE 1 (fp32): dlib's mask as arithmetic -- (x - lo) * 2^100 and (hi - x) * 2^100 are zero exactly on
    // the bound and beyond every admissible eps off it; |med3(df, -g_hi, g_lo)| is dlib's masked |df|
    // wherever that is below eps.  MODE 2 reads the mask off the projected step (mpc_ub_model.h).
    constexpr bool MOVED = MODE == 2;
    T huge = (T)0x1p100;
    asm volatile("" : "+v"(huge));

    ub::Unit<T, EQB> m;
    ub_set_uniform(m, gs, g);
    m.a = m.c = m.as1 = m.cs0 = m.cs1 = m.dlt = m.z0 = m.q1th = (T)0;
    T x[2 * H];
    T x0_prev[2] = {(T)0, (T)0};
    T il[2] = {(T)0, (T)0}, beta = (T)0;
    int64_t k = 0;
    uint32_t iter = 0;
    bool have = false, exhausted = false;
    uint32_t flags = 0;
    uint32_t wave_iters = 0, refills = 0;
#pragma unroll
    for (int q = 0; q < 2 * H; ++q) { x[q] = (T)0; v_put(q, (T)0); }
#pragma unroll
    for (int i = 0; i < H; ++i) w_put(i, (T)0, (T)0);

    auto publish = [&](T a0, T a1, uint32_t it) {
        ((T*)g.front)[k] = m.control(0, a0);
        ((T*)g.rear)[k] = m.control(1, a1);
        if (g.iters) g.iters[k] = (int32_t)it;
    };

    // The queue (longest first) is handed out in two parts.  The first kDeal percent is DEALT: wavefront w of W
    // owns entries w, w + W, w + 2W, ... (every wavefront gets one instance of each block of W neighbours in the order,
    // so the shares are even), knows its next 64 entries a pass ahead (one coalesced-in-time load per lane, issued with
    // the previous pass's record loads) and needs no atomic: a refill pass is ONE memory round trip -- the records --
    // instead of three dependent ones (ticket, queue entry, record: ~5.5 us of which the ticket atomic, at 17 M/s on one
    // address, was the longest).  The rest -- the shortest instances -- goes through the ticket as before and evens out
    // what the deal left uneven.
    // Measured (PG kernel, 262 144 instances, fp64; dealt share 0 / 70 / 85 / 95 %): N = 4: 0.198 / 0.163 / 0.160 / 0.154 ms,
    // N = 10: 0.910 / 0.915 / 0.901 / 0.888, N = 20: 5.34 / 5.41 / 5.40 / 5.52 -- at N = 20 a pass is dominated by the 42 scattered
    // record loads and the set-up behind them, not by the round trips in front, and the deal's 0.6 % of extra wave
    // iterations (shares are even by rank, not by iteration count) cost more than it saves: dealt up to N = 10 only.
    constexpr int kDeal = H <= 10 ? 95 : 0;
    const int wl = lane & (kWave - 1);
    const uint32_t n_waves = gridDim.x * (uint32_t)P::occ;
    const uint32_t wave_id = blockIdx.x * (uint32_t)P::occ + ((uint32_t)threadIdx.x >> 6);
    const uint32_t per_wave = (uint32_t)((n_queue * kDeal / 100) / (int64_t)n_waves);   // dealt entries per wavefront
    const uint32_t dyn_base = per_wave * n_waves;                                            // first entry of the ticket part
    uint32_t dealt = 0;                                                                      // (wave-uniform)
    uint32_t next_k = (uint32_t)wl < per_wave ? order[(int64_t)wl * n_waves + wave_id] : 0u;

#pragma unroll 1
    while (true) {
        // ---- refill: see lane_pg_fused_kernel
        const unsigned long long want = __ballot(!have && !exhausted);
        if (want != 0ull && (__popcll(want) >= UbRefillBatch<H>::value || __ballot(have) == 0ull)) {
            ++refills;
            const uint32_t cnt = (uint32_t)__popcll(want);
            const uint32_t rank = (uint32_t)__popcll(want & ((1ull << wl) - 1ull));
            // the dealt part of the queue first (no atomic, its entries already here), then tickets
            const uint32_t left = per_wave - dealt;
            const uint32_t n_stat = cnt < left ? cnt : left, n_dyn = cnt - n_stat;      // (wave-uniform)
            const uint32_t k_dealt = (uint32_t)__shfl((int)next_k, (int)rank);         // entry dealt + rank sits in lane `rank`
            if (n_stat != 0u) {   // the entries of the next pass: this load travels with the record loads below
                dealt += n_stat;
                next_k = dealt + (uint32_t)wl < per_wave ? order[(int64_t)(dealt + (uint32_t)wl) * n_waves + wave_id] : 0u;
            }
            uint32_t first_ticket = 0;
            if (n_dyn != 0u) {
                if (wl == __ffsll((long long)want) - 1) first_ticket = atomicAdd(ticket, n_dyn);
                first_ticket = (uint32_t)__shfl((int)first_ticket, __ffsll((long long)want) - 1);
            }
            if (!have && !exhausted) {
                const uint32_t t = dyn_base + first_ticket + (rank - n_stat);          // (used by the lanes past the dealt ones)
                if (rank >= n_stat && (int64_t)t >= n_queue) {
                    exhausted = true;
                } else {
                    k = rank < n_stat ? (int64_t)k_dealt : (int64_t)order[t];
                    const T* rec = recs + k * RL;
                    const T* ex = rec + LaneRec<T, H>::kExtra;   // (left by ub_cd_kernel)
#pragma unroll
                    for (int q = 0; q < 2 * H; ++q) x[q] = rec[q];
                    const uint64_t meta = load_meta<T>(rec + 2 * H + 1);
                    // The step constants (two divisions and a square root) and the model scalars (one division) come from
                    // the record, where ub_cd_kernel left them, instead of being recomputed by the few lanes of a pass.
                    // Measured (PG kernel, 262 144 instances, recomputed / from the record): fp64 N = 4: 0.155 / 0.148 ms,
                    // N = 10: 0.887 / 0.875, N = 40: 52.6 / 52.4, fp32 N = 20: 5.35 / 5.26 -- and fp64 N = 20: 5.45 / 5.57 (three
                    // interleaved rounds; its pass is bounded by three dependent memory round trips, which the divisions
                    // used to fill), so that one kernel keeps recomputing.
                    constexpr bool kFromRecord = !(sizeof(T) == 8 && H == 20);
                    iter = (uint32_t)meta;
                    if (meta & kMetaNonFinite) flags |= 0x1u;
                    const bool vinit = (meta & kMetaVInit) != 0;   // mpc.h:330-334, else a fresh v = 0
#pragma unroll
                    for (int q = 0; q < 2 * H; ++q) v_put(q, vinit ? x[q] : m.xz(q & 1));
                    if constexpr (kFromRecord) m.set_instance_ac(ex[3], ex[4], ex[5], ex[6]);
                    else m.set_instance((T)g.step, (T)g.wheelbase, ((const T*)g.v)[k], ((const T*)g.dy)[k], ((const T*)g.dphi)[k]);
                    if ((meta & kMetaStopped) || iter >= kn.max_iter) {
                        // (the coordinate-descent kernel publishes these itself; kept for a queue that holds one)
                        if (!(meta & kMetaStopped)) flags |= 0x2u;
                        if (meta & kMetaNonFinite) { ((T*)g.front)[k] = (T)0; ((T*)g.rear)[k] = (T)0; if (g.iters) g.iters[k] = (int32_t)iter; }
                        else publish(x[0], x[1], iter);
                    } else {
                        if constexpr (kFromRecord) { il[0] = ex[0]; il[1] = ex[1]; beta = ex[2]; }   // mpc.h:342-343 (ub::pg_constants)
                        else ub::pg_constants<T>(rec[2 * H], m.s0, m.s1, il[0], il[1], beta);
                        have = true;
                    }
                }
            }
        }
        if (__ballot(have) == 0ull) {
            if (__ballot(!exhausted) == 0ull) break;
            continue;
        }

        bool stop = false, cap = false;
#pragma unroll 1
        do {
        // v of the LDS-resident steps is fetched one step ahead into a two-slot ring
        T pv[2][2];
        if constexpr (H - 1 >= KV) {
#pragma unroll
            for (int j = 0; j < 2; ++j) pv[(H - 1) & 1][j] = v_get(2 * (H - 1) + j);
        }
        // ---- forward pass
        T Z, Y;
        m.fwd_init(Z, Y);
#pragma unroll
        for (int i = 0; i < H; ++i) {
            m.fwd(Z, Y, x[2 * i], x[2 * i + 1]);
            w_put(i, Z, Y);
        }
        // ---- backward pass fused with the stop test and the speculative update
        x0_prev[0] = x[0]; x0_prev[1] = x[1];
        constexpr int NA = 4;   // accumulators of the stop test (2 / 4 / 8 measured: 5.52 / 5.35 / 5.44 ms at N = 20)
        T acc[NA];
#pragma unroll
        for (int z = 0; z < NA; ++z) acc[z] = (T)0;
        T n0, n1;
        m.bwd_last(n0, n1, Z, Y);
        static_for<H>([&](auto ic) {
            constexpr int i = H - 1 - decltype(ic)::value;
            constexpr int cur = i & 1, nxt = (i - 1) & 1;
            if constexpr (i > 0 && i - 1 >= KV) {
                static_for<2>([&](auto jc) {
                    constexpr int j = decltype(jc)::value;
                    pv[nxt][j] = v_get(2 * (i - 1) + j);
                });
            }
            // pins each step's prefetch of v at the top of its step.  Without it the compiler hoists EVERY step's
            // LDS read to the top of the iteration -- harmless with four steps in LDS (N = 20: 5 % faster without),
            // ~700 register moves per iteration with thirty or forty
            if constexpr (P::step_barrier) __builtin_amdgcn_sched_barrier(0);
            if constexpr (RV) {
                if constexpr (i < H - 1) m.bwd(n0, n1, Z, Y);   // (Z, Y) hold step i (regenerated below)
            } else {
                if constexpr (i < H - 1) m.bwd(n0, n1, w_getz(i), w_gety(i));
            }
            // RV: (Z, Y) of step i-1 from those of step i and x[i] -- BEFORE x[i] is updated below, so that the old
            // x[i] is dead when the new one is defined and the two can share a register (no copy at the back edge)
            if constexpr (RV && i > 0) {
                m.rev(Z, Y, x[2 * i], x[2 * i + 1]);
                asm volatile("" : "+v"(Z), "+v"(Y));   // (keeps the regeneration here: LLVM would sink it past the update)
            }
            T vn[2], st[2];
            static_for<2>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                constexpr int q = 2 * i + j;
                const T xx = x[q];
                T dd;
                if constexpr (j == 0) dd = m.df0(n1, xx); else dd = m.df1(n0, n1, xx);
                vn[j] = m.template project<FAST>(ub::fma_(-il[j], dd, xx), j);      // mpc.h:342
                if constexpr (MOVED) {
                    acc[(2 * i + j) % NA] = tmax(acc[(2 * i + j) % NA], tmin(tabs(dd), tabs(xx - vn[j])));
                } else if constexpr (FAST) {
                    const T g_lo = m.gap_lo(j, xx, huge);
                    const T g_hi = m.gap_hi(j, xx, huge);
                    st[j] = (T)med3_neglo((float)dd, (float)g_hi, (float)g_lo);
                    if constexpr (j == 1) acc[i % NA] = (T)max3_abs((float)acc[i % NA], (float)st[0], (float)st[1]);
                } else {
                    const T up = (xx <= m.bl(j)) ? (T)0 : dd;                       // mpc.h:298-299
                    const T dn = (xx >= m.bh(j)) ? (T)0 : -dd;
                    acc[(2 * i + j) % NA] = tmax(acc[(2 * i + j) % NA], tmax(up, dn));
                }
            });
            static_for<2>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                constexpr int q = 2 * i + j;
                T vold;
                if constexpr (i >= KV) vold = pv[cur][j]; else vold = r_v[q];
                x[q] = m.template project<FAST>(ub::fma_(beta, vn[j] - vold, vn[j]), j);   // mpc.h:343 (difference form: pg_update)
                if constexpr (i < KV) r_v[q] = vn[j];
            });
            if constexpr (i >= KV) {
                static_for<2>([&](auto jc) {   // adjacent stores: one ds_write2st64 per step
                    constexpr int j = decltype(jc)::value;
                    v_put(2 * i + j, vn[j]);
                });
            }
        });
        T max_df = acc[0];
#pragma unroll
        for (int z = 1; z < NA; ++z) max_df = tmax(max_df, acc[z]);
        ++wave_iters;
        stop = have && (max_df < geps);                                         // mpc.h:310-311
        ++iter;
        cap = have && !stop && iter >= kn.max_iter;                             // mpc.h:271
        if (__ballot(stop || cap) != 0ull) {
            if (stop) {
                publish(x0_prev[0], x0_prev[1], iter - 1);
                have = false;
            }
            if (cap) {
                flags |= 0x2u;
                publish(x[0], x[1], iter);
                have = false;
            }
            const unsigned long long waiting = __ballot(!have && !exhausted);
            if (__popcll(waiting) >= UbRefillBatch<H>::value || __ballot(have) == 0ull) break;
        }
        } while (true);
    }
    raise_flags(g.flags, flags);
    if (stats && (lane & (kWave - 1)) == 0) {
        atomicAdd(&stats[0], (unsigned long long)wave_iters);
        atomicAdd(&stats[1], (unsigned long long)refills);
    }
}

}  // namespace tpc
