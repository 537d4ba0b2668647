// LANE kernels: one lane per MPC instance, 64 instances per wavefront.
//
// A batch is three launches, all bit-faithful to dlib::mpc::solve_linear_mpc (mpc.h:253-347):
//
//   lane_cd_kernel        coordinate-descent phase (iter < smo_iters, mpc.h:319-335).  Every lane
//                         of a wave runs the same <= smo_iters iterations in lockstep, so there is
//                         no divergence to manage.  Q_diag and MM sit in LDS ([var][lane],
//                         conflict-free, lane-private: no barriers); Q_diag is read at a per-lane
//                         dynamic index.  Leaves {controls, lambda, iter} of each instance in a
//                         per-instance record in HBM plus a queue key (the bits of float(lambda)).
//   (mpc_sort.hip)        orders the instances longest-first by that key.
//   lane_pg_fused_kernel  accelerated projected-gradient phase (mpc.h:336-345).  Iteration counts
//                         differ 30x between instances, so waves are
//                         persistent and a lane whose instance has converged pulls the next one
//                         from an atomic ticket over the sorted queue: no lane waits for the
//                         slowest instance of its wave.  See the comment on the kernel.
//   lane_pg_kernel        the same phase, unfused, for callers that want the controller state
//                         (all controls and dlib's v) back: warm-start chains and tpc_mpc_rollout.
//
// Built for one wave per SIMD at fp64: there every instruction costs one ~2.02 ns issue slot
// (scripts/ubench_clock.hip), so the kernels are shaped by instruction count, not by latency hiding
// through occupancy.
#pragma once

#include <type_traits>
#include <utility>

#include "mpc_model.h"

namespace tpc {

// Per-instance record handed from the CD phase to the PG phase:
//   rec[0 .. 2H-1] controls (indexed 2*i + j), rec[2H] lambda, rec[2H+1] meta (bit pattern), then kExtras values the
//   LANE_FMA / GROUP kernels leave for their refill passes (mpc_ub.h: step constants, model scalars, target)
// meta: low 32 bits iteration count; bit 32 = stopped (eps reached); bit 33 = v := u was executed
// at the last CD iteration (mpc.h:330-334); bit 34 = non-finite inputs; bit 35 = invalid model.
constexpr uint64_t kMetaStopped = 1ull << 32;
constexpr uint64_t kMetaVInit = 1ull << 33;
constexpr uint64_t kMetaNonFinite = 1ull << 34;
constexpr uint64_t kMetaBadModel = 1ull << 35;   // violates dlib's requires clause: not solved

template <typename T, int H> struct LaneRec {
    // record length in T elements, padded to an even count of 8-byte words
    static constexpr int kMetaT = sizeof(T) == 8 ? 1 : 2;         // meta needs 64 bits
    static constexpr int kExtra = 2 * H + 1 + kMetaT;              // first extra value
    static constexpr int kExtras = 7;
    static constexpr int kLen = ((2 * H + 1 + kMetaT + kExtras) + 1) / 2 * 2;
};

template <typename T> TPC_DEV void store_meta(T* rec, uint64_t meta);
template <> TPC_DEV void store_meta<double>(double* rec, uint64_t meta) {
    rec[0] = __longlong_as_double((long long)meta);
}
template <> TPC_DEV void store_meta<float>(float* rec, uint64_t meta) {
    rec[0] = __uint_as_float((uint32_t)meta);
    rec[1] = __uint_as_float((uint32_t)(meta >> 32));
}
template <typename T> TPC_DEV uint64_t load_meta(const T* rec);
template <> TPC_DEV uint64_t load_meta<double>(const double* rec) {
    return (uint64_t)__double_as_longlong(rec[0]);
}
template <> TPC_DEV uint64_t load_meta<float>(const float* rec) {
    return (uint64_t)__float_as_uint(rec[0]) | ((uint64_t)__float_as_uint(rec[1]) << 32);
}

// Output plumbing shared by both model kinds.
template <typename T, int I, int H, class Args> struct LaneIO;

template <typename T, int I, int H> struct LaneIO<T, I, H, CompactArgs> {
    static TPC_DEV void init_controls(const CompactArgs&, int64_t, T* u) {
#pragma unroll
        for (int i = 0; i < 2 * H; ++i) u[i] = (T)0;
    }
    template <class VSet> static TPC_DEV void load_v(const CompactArgs&, int64_t, VSet vset) {
#pragma unroll
        for (int i = 0; i < 2 * H; ++i) vset(i, (T)0);
    }
    template <class VGet>
    static TPC_DEV void write(const CompactArgs& g, int64_t k, const T* u, VGet, uint32_t it) {
        ((T*)g.front)[k] = u[0];
        ((T*)g.rear)[k] = u[1];
        if (g.iters) g.iters[k] = (int32_t)it;
    }
};

template <typename T, int I, int H> struct LaneIO<T, I, H, GeneralArgs> {
    static TPC_DEV void init_controls(const GeneralArgs& g, int64_t k, T* u) {
#pragma unroll
        for (int i = 0; i < 2 * H; ++i) u[i] = (T)0;
        if (g.controls) {
            const T* cp = (const T*)g.controls + k;
            // warm-start shift (mpc.h:231-232): controls[i-1] = controls[i], last one kept
#pragma unroll
            for (int i = 0; i < H; ++i)
#pragma unroll
                for (int j = 0; j < I; ++j) {
                    const int src = (g.shift_controls && i + 1 < H) ? i + 1 : i;
                    u[2 * i + j] = cp[(int64_t)(src * I + j) * g.ld];
                }
        }
    }
    template <class VSet> static TPC_DEV void load_v(const GeneralArgs& g, int64_t k, VSet vset) {
        const T* vp = (const T*)g.v + k;
#pragma unroll
        for (int i = 0; i < H; ++i)
#pragma unroll
            for (int j = 0; j < I; ++j)
                vset(2 * i + j, g.v ? vp[(int64_t)(i * I + j) * g.ld] : (T)0);
    }
    template <class VGet>
    static TPC_DEV void write(const GeneralArgs& g, int64_t k, const T* u, VGet vget, uint32_t it) {
#pragma unroll
        for (int j = 0; j < I; ++j) ((T*)g.u0)[(int64_t)j * g.ld + k] = u[j];
        if (g.controls) {
            T* cp = (T*)g.controls + k;
#pragma unroll
            for (int i = 0; i < H; ++i)
#pragma unroll
                for (int j = 0; j < I; ++j) cp[(int64_t)(i * I + j) * g.ld] = u[2 * i + j];
        }
        if (g.v) {
            T* vp = (T*)g.v + k;
#pragma unroll
            for (int i = 0; i < H; ++i)
#pragma unroll
                for (int j = 0; j < I; ++j) vp[(int64_t)(i * I + j) * g.ld] = vget(2 * i + j);
        }
        if (g.iters) g.iters[k] = (int32_t)it;
    }
};

// ------------------------------------------------------------------------------------------------
// Phase 1: coordinate descent.  grid = ceil(n/64) blocks of one wave.
// Waves per SIMD the CD kernel is compiled for: small horizons need few registers and little LDS,
// and with several waves per SIMD the 32-bit select chains issue at their 2-cycle rate.
template <typename T, int H> struct CdOcc {
    static constexpr int value = H * (int)sizeof(T) <= 40 ? 4 : (H * (int)sizeof(T) <= 80 ? 2 : 1);
};

//
// RESOLVE = true restricts the kernel to the instances of a batch that a tolerance family (WAVE, LANE_FMA, GROUP) left
// on the iteration cap: `select[k] == max_iter` picks them, `gate` (the batch's flag word) says whether there is any,
// and instead of a sort key each picked instance is appended to the queue the fused projected-gradient kernel reads
// (`keys` = the queue, `key_hist` = its length; the order is irrelevant: they all run max_iter iterations).  Those
// instances have not converged, and over thousands of iterations of an ill-conditioned problem the tolerance
// families' rounding differences grow (2.3e-5 seen under adversarial parameters, profiles/r03_fuzz_lane_fma.txt): AUTO
// solves them once more here and in lane_pg_fused_kernel, in dlib's own operation order, and publishes dlib's bits
// (tpc_mpc_api.cpp, wants_cap_resolve).  With the gate closed -- the usual case -- every wavefront leaves at once, the
// queue stays empty and the projected-gradient kernels return on their first look at it.
//
// SUBSET 2 (PRESOLVE) picks by PREDICTION instead: the instances whose lambda (mpc.h:116-123) is at least `lambda_from` --
// the iteration count grows like 5.5 sqrt(lambda), so lambda >= (max_iter / 7)^2 names the instances that will end on
// the cap (tpc_mpc_api.cpp, presolve_lambda) -- are queued BEFORE any tolerance family has run, so that their 10 000
// bit-exact iterations run beside that family's pass instead of behind it.  SUBSET 1 then skips what SUBSET 2 took
// (the same comparison on the same lambda).
template <typename T, int I, int H, class Model, class Args, int SUBSET = 0>
__global__ __launch_bounds__(64, (CdOcc<T, H>::value)) void lane_cd_kernel(Args g, Knobs kn, T* __restrict__ recs,
                                                         uint32_t* __restrict__ keys,
                                                         uint32_t* __restrict__ key_rank,
                                                         uint32_t* __restrict__ key_hist,
                                                         unsigned long long* __restrict__ stats,
                                                         int publish_finished, const int32_t* __restrict__ select = nullptr,
                                                         const uint32_t* __restrict__ gate = nullptr, T lambda_from = (T)0,
                                                         const uint32_t* __restrict__ pre_len = nullptr, uint32_t pre_limit = 0u) {
    constexpr bool RESOLVE = SUBSET != 0;
    constexpr int RL = LaneRec<T, H>::kLen;
    __shared__ T s_qd[2 * H][kWave];   // Q_diag[i](j) of lane l at s_qd[2*i + j][l]
    __shared__ T s_mm[2 * H][kWave];   // MM[i](j)
    const int lane = threadIdx.x;
    const int64_t k = (int64_t)blockIdx.x * kWave + lane;
    if (k >= g.n) return;   // no barriers below: a partial last wave just runs with fewer lanes
    if constexpr (SUBSET == 1) {
        if ((__builtin_nontemporal_load(gate) & 0x2u) == 0u) return;      // nothing ended on the cap: the usual case
        if (select[k] != (int32_t)kn.max_iter) return;
    }

    Model m;
    m.load(g, k);
    const bool nonfinite = m.nonfinite();
    const bool badmodel = m.invalid();

    T u[2 * H], w[2 * H];
    LaneIO<T, I, H, Args>::init_controls(g, k, u);
    const T lambda = ctor_lambda_qdiag<T, I, H>(m, [&](int i, int j, T val) { s_qd[2 * i + j][lane] = val; });
    if constexpr (SUBSET == 1) {   // solved ahead by SUBSET 2 -- unless that queue grew past what its kernel takes (mpc_lanex.h, SOLO)
        if (lambda_from > (T)0 && lambda >= lambda_from && __builtin_nontemporal_load(pre_len) <= pre_limit) return;
    }
    if constexpr (SUBSET == 2) { if (!(lambda >= lambda_from)) return; }
    T mm_max = (T)0;
    bool mm_nan = false;   // fmax drops a NaN operand: a NaN element of the linear term is tracked on its own
    linear_term<T, I, H>(m, w, [&](int q, T val) { s_mm[q][lane] = val; mm_max = tmax(mm_max, tabs(val)); mm_nan = mm_nan || val != val; });
    if (mm_nan) mm_max = (T)__builtin_inf();   // beyond every screen

    const T eps = (T)kn.eps;
    // one instance outside the screen sends the whole batch to the exact-stop-test build of the
    // fused PG kernel (stats[2], read by both builds at launch; see lane_pg_fused_kernel)
    // (one atomic per wavefront at most: when the failing condition is batch-wide -- bounds that do
    // not straddle zero, a huge eps -- every lane fails, and n atomics on one word would serialise)
    if constexpr (Model::kFastStop && !RESOLVE) {
        const unsigned long long failing = __ballot(!m.fast_stop_ok(mm_max, eps, lambda, H));
        if (failing != 0ull && lane == __ffsll((long long)failing) - 1) atomicOr(&stats[2], 1ull);
    }
    // (RESOLVE: the exact-stop-test build of the fused kernel is the only one launched behind this kernel -- one launch
    // fewer on every call that has nothing to re-solve -- so every wavefront that got here asks for it)
    if constexpr (RESOLVE) {
        const unsigned long long here = __ballot(true);
        if (lane == __ffsll((long long)here) - 1) atomicOr(&stats[2], 1ull);
    }
    const uint32_t cd_iters = kn.smo_iters < kn.max_iter ? kn.smo_iters : kn.max_iter;
    uint32_t iter = 0;
    bool stopped = (Model::kScreen && nonfinite) || badmodel;   // see CompactModel::kScreen, GeneralModel::invalid
    bool vinit = false;
    // Two builds of the arg-max scan (the arithmetic of the gradient and of the step is dlib's in both, so both are
    // bit-exact; only how the SAME decisions are reached differs -- see ub_cd_kernel, mpc_ub.h):
    //   exact   dlib's mask by compare and select (mpc.h:298-299), running maximum, its index, sign and u;
    //   fast    where every lane of the wavefront passed the screen of the select-free stop test and starts inside
    //           its box (a cold start; a caller's warm start may lie outside): the mask as arithmetic on the gaps
    //           (u - lower) 2^600 and (upper - u) 2^600, which are 0 exactly on the bound and beyond every |df| off it,
    //           and ONE signed value for the running arg-max.
    // (the winner's u is tracked by select: an LDS copy of u read by index -- ub_cd_kernel's way -- would be a third
    // array beside s_qd and s_mm and cost residency at every horizon: measured at N = 20, 1.0 ms against 0.68)
    constexpr T kHuge = (T)(sizeof(T) == 8 ? 0x1p600 : 0x1p100);
    T huge = kHuge;
    asm volatile("" : "+v"(huge));
    T lo_h[I], hi_h[I];
#pragma unroll
    for (int j = 0; j < I; ++j) { lo_h[j] = -(m.lo(j) * huge); hi_h[j] = m.hi(j) * huge; }

    auto iteration = [&](auto fast_tag, uint32_t it) {
        constexpr bool F = decltype(fast_tag)::value;
        gradient<T, I, H>(m, u, [&](int q) { return s_mm[q][lane]; }, w);
        // arg-max |df| over free variables, scanning i then j, strict '>' (mpc.h:289-309)
        T max_df = (T)0, best_u = (T)0, best_mm = (T)0;
        int best = 0, best_sign = 0;
#pragma unroll
        for (int i = 0; i < H; ++i)
#pragma unroll
            for (int j = 0; j < I; ++j) {
                const int q = 2 * i + j;
                const T uu = u[q], dd = w[q];
                if constexpr (F) {
                    const T g_lo = tfma(uu, huge, lo_h[j]), g_hi = tfma(-huge, uu, hi_h[j]);
                    const T mm = tmax(tmin(dd, g_lo), -g_hi);           // df where dlib counts it, else 0
                    const bool better = tabs(mm) > tabs(best_mm);
                    best_mm = better ? mm : best_mm;
                    best = better ? q : best;
                    best_u = better ? uu : best_u;
                } else {
                    // select form of `if (!blocked && |df| > max_df)`: a variable at its lower bound may
                    // only contribute a negative df, one at its upper bound a positive one; the gated
                    // magnitude is |df| or 0, NaN stays NaN and loses every `>` like in dlib.  (As an
                    // `if` the compiler built a branch per variable: 1430 instructions per iteration.)
                    const T up = (uu <= m.lo(j)) ? (T)0 : dd;
                    const T dn = (uu >= m.hi(j)) ? (T)0 : -dd;
                    const T mag = tmax(up, dn);
                    const bool better = mag > max_df;
                    max_df = tmax(max_df, mag);               // == better ? mag : max_df, NaN included
                    best = better ? q : best;
                    best_sign = better ? sign_word(dd) : best_sign;   // df[best] = +-max_df: only its sign is kept
                    best_u = better ? uu : best_u;
                }
            }
        if constexpr (F) max_df = tabs(best_mm);
        // (select form, no divergent block: a conditional update of the register array costs a copy of all of it)
        stopped = stopped || max_df < eps;                      // mpc.h:310-311
        const bool act = !stopped;
        T best_df;
        if constexpr (F) best_df = best_mm; else best_df = with_sign(max_df, best_sign);
        const T qdv = s_qd[best][lane];
        const bool upd = act && qdv != (T)0;                    // mpc.h:322 (`continue` still counts)
        // (bounds picked by select: a run-time index would put the model in scratch)
        const bool second = I == 2 && (best & 1);
        const T blo = second ? m.lo(I - 1) : m.lo(0), bhi = second ? m.hi(I - 1) : m.hi(0);
        T nu = -(best_df - qdv * best_u) / (upd ? qdv : (T)1);  // mpc.h:325
        nu = put_in_range(blo, bhi, nu);                        // mpc.h:326
        const int sel = upd ? best : -1;
#pragma unroll
        for (int q = 0; q < 2 * H; ++q)
            if ((q & 1) < I) u[q] = (q == sel) ? nu : u[q];
        vinit = upd ? (it + 1 == kn.smo_iters) : vinit;         // mpc.h:330-334
        iter += act ? 1u : 0u;
    };
    bool fast_cd = false;
    if constexpr (Model::kFastStop) {
        bool inside = true;
#pragma unroll
        for (int q = 0; q < 2 * H; ++q)
            if ((q & 1) < I) inside = inside && u[q] >= m.lo(q & 1) && u[q] <= m.hi(q & 1);
        fast_cd = __ballot(!(m.fast_stop_ok(mm_max, eps, lambda, H) && inside)) == 0ull;
    }
    if (fast_cd) {
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
    for (int q = 0; q < 2 * H; ++q) rec[q] = ((q & 1) < I) ? u[q] : (T)0;
    rec[2 * H] = lambda;
    uint64_t meta = (uint64_t)iter;
    if (stopped) meta |= kMetaStopped;
    if (vinit) meta |= kMetaVInit;
    if (nonfinite) meta |= kMetaNonFinite;
    if (badmodel) meta |= kMetaBadModel;
    store_meta<T>(rec + 2 * H + 1, meta);
    // queue key: instances with the largest lambda need the most projected-gradient iterations
    // (rank correlation 0.97 on the synthetic workload); a caller that knows better -- typically
    // the iteration counts the same slots needed in the previous control cycle -- passes them as
    // work_hint and gets a queue in that order instead.  The key never influences a result.
    const bool finished = stopped || iter >= kn.max_iter;
    // Where lambda sits near its floor H*sum(R) (the trans(M)*Q*M part adds < 50 %) it says nothing
    // about the iteration count (rank correlation 0.07 in that range: slow vehicles, whose few
    // hundred iterations vary 7x); left at the end of the queue such instances decide when the
    // kernel ends.  They get the top key instead and start first, where their spread is absorbed
    // (simulated and measured: -6 % kernel time at H = 20, -20 % at H = 10 with the reference's
    // weights; no effect on workloads whose lambda stays off the floor).
    T lambda_floor = m.R(0);
    if (I == 2) lambda_floor = lambda_floor + m.R(I - 1);
    lambda_floor = lambda_floor * (T)H;
    const bool uninformative = lambda < (T)1.5 * lambda_floor;
    const float lf = g.work_hint ? (float)(g.work_hint[k] > 0 ? g.work_hint[k] : 1) : (float)lambda;
    // The two classes whose internal order does not matter -- finished instances (bottom of the
    // queue) and uninformative ones (top) -- are spread over 128 bins each by their lane index:
    // one bin each meant up to 10^5 atomics queueing on a single address (most of the CD kernel's
    // time at H = 4, where both classes are large).
    const uint32_t spread = (uint32_t)k & 127u;
    uint32_t key = __float_as_uint(lf);
    if (finished) key = spread << 16;                                             // bins 0..127: below every positive normal float
    else if (!g.work_hint && uninformative) key = (0x7f00u + spread) << 16;       // above every lambda
    else if (!(lf > 0.0f) || key < 0x00800000u) key = 0x00800000u;                // (an absurd lambda stays between them)
    else if (key >= 0x7f000000u) key = 0x7effffffu;
    // When the caller does not want the controller state back, an instance this phase finished
    // (every instance when max_iter <= smo_iters, under 1 % of the synthetic workload otherwise) is
    // complete: its outputs go out here, and the PG kernel's queue
    // ends where bins 0..127 begin (the scan leaves that position in the histogram, mpc_sort.hip).
    uint32_t f = 0;
    if (publish_finished && finished) {
        LaneIO<T, I, H, Args>::write(g, k, u, [&](int) { return (T)0; }, iter);
        if (nonfinite) f |= 0x1u;
        if (badmodel) f |= 0x4u;
        if (!stopped) f |= 0x2u;          // ran into max_iter inside this phase
    }
    if constexpr (RESOLVE) {
        (void)key;
        if (!finished) keys[atomicAdd(key_hist, 1u)] = (uint32_t)k;   // the queue of lane_pg_fused_kernel, in any order
        return;
    }
    raise_flags(g.flags, f);              // (every lane of the wavefront gets here)
    keys[k] = key;
    key_rank[k] = atomicAdd(&key_hist[key >> 16], 1u);   // counting sort: histogram + rank in bin (mpc_sort.hip)
}

// ------------------------------------------------------------------------------------------------
// Phase 2: accelerated projected gradient with lane refill.  Persistent waves; grid <= waves the
// chip holds.  `ticket` must be zero at launch.
template <typename T, int I, int H, class Model, class Args>
__global__ __launch_bounds__(64, 1) void lane_pg_kernel(Args g, Knobs kn, const T* __restrict__ recs,
                                                         const uint32_t* __restrict__ order,
                                                         uint32_t* __restrict__ ticket,
                                                         unsigned long long* __restrict__ stats, int only_if_refused = 0) {
    constexpr int RL = LaneRec<T, H>::kLen;
    // (behind the general-form GROUP kernel, mpc_groupg_inst.hip: only for a batch the stop-test screen refused)
    if (only_if_refused && __builtin_nontemporal_load(&stats[2]) == 0ull) return;
    __shared__ T s_mm[2 * H][kWave];   // MM[i](j) of lane l at s_mm[2*i + j][l]
    __shared__ T s_v[2 * H][kWave];    // v[i](j)   (mpc.h:250)
    const int lane = threadIdx.x;
    const T eps = (T)kn.eps;
    auto vget = [&](int q) { return s_v[q][lane]; };
    auto vset = [&](int q, T val) { s_v[q][lane] = val; };

    Model m;
    T u[2 * H], w[2 * H];
    T inv_lambda = (T)0, beta = (T)0;
    int64_t k = 0;
    uint32_t iter = 0;
    bool have = false;        // this lane holds an unfinished instance
    bool exhausted = false;   // the ticket ran past n: nothing left to pull
    uint32_t flags = 0;
    uint32_t wave_iters = 0, refills = 0;   // occupancy statistics (two atomics per wave at exit)

#pragma unroll 1
    while (true) {
        // ---- refill (rare, wave-uniform branch): free lanes pull the next instance; instances the
        // CD phase already finished are written out on the spot
        if (__ballot(!have && !exhausted) != 0ull) {
            ++refills;
            if (!have && !exhausted) {
                const uint32_t t = atomicAdd(ticket, 1u);
                if ((int64_t)t >= g.n) {
                    exhausted = true;
                } else {
                    k = (int64_t)order[t];   // longest-first queue
                    const T* rec = recs + k * RL;
#pragma unroll
                    for (int q = 0; q < 2 * H; ++q)
                        if ((q & 1) < I) u[q] = rec[q];
                    const T lambda = rec[2 * H];
                    const uint64_t meta = load_meta<T>(rec + 2 * H + 1);
                    iter = (uint32_t)meta;
                    if (meta & kMetaNonFinite) flags |= 0x1u;
                    if (meta & kMetaBadModel) flags |= 0x4u;
                    if (meta & kMetaVInit) {
#pragma unroll
                        for (int q = 0; q < 2 * H; ++q)
                            if ((q & 1) < I) vset(q, u[q]);
                    } else {
                        LaneIO<T, I, H, Args>::load_v(g, k, vset);
                    }
                    if ((meta & kMetaStopped) || iter >= kn.max_iter) {
                        if (!(meta & kMetaStopped)) flags |= 0x2u;
                        LaneIO<T, I, H, Args>::write(g, k, u, vget, iter);
                    } else {
                        m.load(g, k);
                        linear_term<T, I, H>(m, w, [&](int q, T val) { s_mm[q][lane] = val; });
                        inv_lambda = (T)1.0 / lambda;                         // mpc.h:342
                        const T sq = tsqrt(lambda);
                        beta = (sq - (T)1) / (sq + (T)1);                     // mpc.h:343
                        have = true;
                    }
                }
            }
        }
        if (__ballot(have) == 0ull) {
            if (__ballot(!exhausted) == 0ull) break;   // every lane saw the end of the queue
            continue;                                  // drew only finished instances: pull again
        }

        // ---- one solver iteration on all 64 lanes (idle lanes compute on stale registers; their
        // results are never stored).  Straight-line code: gradient, stop test, update.
        gradient<T, I, H>(m, u, [&](int q) { return s_mm[q][lane]; }, w);
        // largest free gradient component (mpc.h:289-309); the arg-max index is not needed here.
        // max() is exact, so four independent accumulators give the same value as dlib's scan.
        T acc[4] = {(T)0, (T)0, (T)0, (T)0};
#pragma unroll
        for (int i = 0; i < H; ++i)
#pragma unroll
            for (int j = 0; j < I; ++j) {
                const T uu = u[2 * i + j], dd = w[2 * i + j];
                const T up = (uu <= m.lo(j)) ? (T)0 : dd;      // at lower bound: only df<0 counts
                const T dn = (uu >= m.hi(j)) ? (T)0 : -dd;     // at upper bound: only df>0 counts
                acc[(i * I + j) & 3] = tmax(acc[(i * I + j) & 3], tmax(up, dn));
            }
        const T max_df = tmax(tmax(acc[0], acc[1]), tmax(acc[2], acc[3]));
        const bool stop = have && (max_df < eps);               // mpc.h:310-311
        if (__ballot(stop) != 0ull) {
            if (stop) {
                LaneIO<T, I, H, Args>::write(g, k, u, vget, iter);
                have = false;
            }
        }
#pragma unroll
        for (int i = 0; i < H; ++i)
#pragma unroll
            for (int j = 0; j < I; ++j) {
                const int q = 2 * i + j;
                const T v_old = vget(q);
                const T vn = clamp3(u[q] - inv_lambda * w[q], m.lo(j), m.hi(j));   // mpc.h:342
                vset(q, vn);
                u[q] = clamp3(vn + beta * (vn - v_old), m.lo(j), m.hi(j));         // mpc.h:343
            }
        ++iter;
        ++wave_iters;
        const bool cap = have && iter >= kn.max_iter;           // mpc.h:271
        if (__ballot(cap) != 0ull) {
            if (cap) {
                flags |= 0x2u;
                LaneIO<T, I, H, Args>::write(g, k, u, vget, iter);
                have = false;
            }
        }
    }
    raise_flags(g.flags, flags);
    if (stats && lane == 0) {
        atomicAdd(&stats[0], (unsigned long long)wave_iters);   // wave-iterations executed
        atomicAdd(&stats[1], (unsigned long long)refills);      // refill blocks executed
    }
}

// compile-time loop: f(std::integral_constant<int, 0>) ... f(std::integral_constant<int, N-1>)
template <class F, int... Is> TPC_DEV void static_for_impl(F&& f, std::integer_sequence<int, Is...>) {
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F> TPC_DEV void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

// ------------------------------------------------------------------------------------------------
// Phase 2, fused and software-pipelined form: the throughput kernel.  Same arithmetic as
// lane_pg_kernel, scheduled for ONE wave per SIMD, where (measured, scripts/ubench_clock.hip,
// ubench_lds2.hip) every VALU or SALU instruction costs one ~2.02 ns issue slot -- 4.83 cycles at the
// 2.34-2.39 GHz the chip holds under fp64 load, the most a lone wave gets out of the fp64 pipe (two
// waves per SIMD reach 4.2-4.4 cycles together) --, an LDS instruction of any width about three
// slots, and nothing hides LDS latency but the wave's own instruction stream:
//   * the backward pass, the stop test and the projected-gradient update are fused per horizon
//     step: as soon as df[i] exists its contribution to max|df| is taken and u[i], v[i] are
//     advanced speculatively.  dlib updates only when the stop test fails; a lane that stops
//     publishes the controls it had BEFORE this update (u[0] is all the caller receives, so only
//     u[0] is kept) and refills, so the speculation is never observable;
//   * MM (read once per iteration) and v (read once, written once) live in LDS as [var][lane]
//     columns (20 + 12 KB per wave at H=20 fp64, where 8 steps of v stay in VGPRs); each step's four
//     values are fetched ONE STEP AHEAD into a small register ring, so their latency hides under
//     the previous step's ~55 instructions; one __builtin_amdgcn_sched_barrier(0) per step pins the
//     prefetch at the top of its step (without it the loads sink to their uses; a second barrier
//     at the end of the step costs 18 instructions per iteration).  u and the forward-pass array w stay in VGPRs.
// Used whenever the caller does not ask for the controller state back (controls_inout / v_inout).
// An AGPR-resident variant (v_accvgpr_read/write instead of LDS) was measured 2.5 % slower: twelve
// moves per step cost more issue slots than three LDS instructions.
// fp32 halves the register footprint: u, w, MM and v all fit the VGPR file (4 x 2H floats), so the
// fp32 kernel uses no LDS at all and two waves share a SIMD, where 32-bit VALU instructions issue
// at their 2-cycle rate instead of one per 4-cycle slot.  (With MM and v in LDS, two waves per SIMD
// gained nothing: eight waves' ds_read2/ds_write kept the CU's one LDS pipe busy 90 % of the time.)
// The same holds for fp64 up to H = 10 (4 x 2H doubles = 160 VGPRs).
// Free lanes wait until this many of them can be refilled by one pass (see the refill block).  The
// pass costs about 4 us whatever the horizon, an iteration 0.2 us at H = 4 and 2.4 us at H = 20, so
// short horizons batch more.  Measured optima (PG kernel, n = 262 144): H = 4: 1.09 ms at 2, 0.27 ms
// at 24-48; H = 5: 0.34 ms at 24; H = 10: flat from 3 to 8; H = 20: 10.24 ms at 2, 9.82 at 3-4.
template <int H> struct RefillBatch { static constexpr int value = H <= 5 ? 24 : (H <= 10 ? 6 : 3); };
// measured per kernel (scripts/ab_many.sh, both forms at every horizon and dtype)
template <typename T, int H> struct ExitEveryStop { static constexpr bool value = (sizeof(T) == 8 && H == 20) || (sizeof(T) == 4 && H == 30); };
// fp32 at H = 30 / 40 keeps everything in registers as well (240 / 320 of them), at one wave per SIMD: without
// the LDS round trips of MM and v the kernels run 29.6 -> 26.0 ms (H = 30) and 73.6 -> 67.4 ms (H = 40) per
// 262 144 instances.
template <typename T, int H> struct FusedInRegs { static constexpr bool value = H * (int)sizeof(T) <= 80 || sizeof(T) == 4; };
#ifdef TPC_KV_STEPS   // A/B override: horizon steps of v kept in VGPRs by the LDS/AGPR plans
template <typename T, int H> struct FusedVRegSteps { static constexpr int value = FusedInRegs<T, H>::value ? 0 : TPC_KV_STEPS; };
#else
template <typename T, int H> struct FusedVRegSteps {
    static constexpr int value = sizeof(T) != 8 ? 0 : (H == 20 ? 6 : ((H == 30 || H == 40) ? 8 : 0));
};
#endif
// (four waves per SIMD for the short fp32 horizons, whose state would allow it, measured 10-12 % SLOWER:
// fewer lanes per refill pass and more waves pulling on the one ticket)
template <typename T, int H> struct FusedOcc { static constexpr int value = H * (int)sizeof(T) <= 80 ? 2 : 1; };   // 160 state registers: two waves per SIMD
// fp64, H = 30: u (120 words) fits the VGPRs, but MM and v in LDS take 61 KB per wave and leave two
// of a CU's four SIMDs without a wave.  The forward-pass array w and all of v go to AGPRs instead
// (240 of the 256; two v_accvgpr moves per double and direction: 550 of the loop's 2130
// instructions), MM alone stays in LDS (31 KB): four waves per CU, 70 -> 43 ms for 262 144
// instances.  At H = 40 the same plan was measured 1.5x SLOWER than MM + v in LDS (80 KB, two
// waves per CU, w spilled by the compiler): only 20 steps of v fit the AGPRs next to w, LDS still
// allows two waves only, and the moves are pure cost.
template <typename T, int H> struct FusedBig { static constexpr bool value = sizeof(T) == 8 && H == 30; };
// fp64, H = 40, compact model: the state of one instance is 4 x 80 doubles = 640 registers' worth,
// against 512 registers per lane plus 160 of LDS per lane when all four SIMDs of a CU hold a wave --
// it does not fit.  Round 1 kept MM and v in LDS (80 KB per wave: two waves per CU, half the SIMDs
// idle) and let the compiler spill w: 10.2 us per iteration.  Here the forward-pass array w is
// CHECKPOINTED instead: only the even steps are kept (in AGPRs), and the backward pass recomputes
// M[i] for odd i from M[i-1] and the not-yet-updated u[i] with the forward pass's own operations
// (bit-identical; the multiplies are opaque to the optimiser, which would otherwise merge the
// re-computation with the original and keep the value alive).  That is 8 extra flops per second
// step, and the state shrinks to u (VGPRs) + 80 words of w + v (VGPRs and AGPRs) + MM (LDS, 40 KB
// per wave): four waves per CU.
// The same plan pays at H = 30 too (compact model): against FusedBig's all-AGPR w and v it halves w, keeps
// 16 steps of v in VGPRs instead of 8 and so saves ~320 of the loop's 550 AGPR moves for ~120 recomputed
// flops: 37.8 -> 35.7 ms per 262 144 instances (KV = 8 / 12 / 16 / 20: 37.3 / 36.1 / 35.7 / 36.8 ms), same bits.
template <typename T, int H> struct FusedCkpt { static constexpr bool value = sizeof(T) == 8 && (H == 40 || H == 30); };
template <typename T, int H> struct FusedCkptVRegSteps { static constexpr int value = H == 30 ? 16 : FusedVRegSteps<T, H>::value; };

// fp32 only: v_med3_f32 is dlib's three-argument clamp in ONE instruction, and v_max3_f32 folds two
// stop-test terms into the running maximum in one -- for operands that cannot be NaN (med3 returns the
// minimum then, the clamp the upper bound), i.e. in the build the model's screen admits.  min / max / med3 /
// max3 issue at the 4-cycle rate where plain fp32 multiplies and adds issue at 2 (scripts/ubench_pk.hip), so
// they were 38 % of the fp32 iteration's time for 27 % of its instructions.  (fp64 has no med3.)
TPC_DEV float med3(float x, float lo, float hi) {
    float r;
    asm("v_med3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(lo), "v"(hi));
    return r;
}
TPC_DEV float med3_neglo(float x, float neg_lo, float hi) {   // clamp of x to [-neg_lo, hi]
    float r;
    asm("v_med3_f32 %0, %1, -%2, %3" : "=v"(r) : "v"(x), "v"(neg_lo), "v"(hi));
    return r;
}
TPC_DEV float max3_abs(float acc, float a, float b) {        // max(acc, |a|, |b|)
    float r;
    asm("v_max3_f32 %0, %1, |%2|, |%3|" : "=v"(r) : "v"(acc), "v"(a), "v"(b));
    return r;
}
template <bool NONAN, typename T> TPC_DEV T clamp3_fast(T val, T lo, T hi) {
    if constexpr (NONAN && sizeof(T) == 4) return (T)med3((float)val, (float)lo, (float)hi);
    else return clamp3(val, lo, hi);
}

template <typename T, int I, int H, class Model, class Args, bool FAST>
__global__ __launch_bounds__((64 * FusedOcc<T, H>::value), (FusedOcc<T, H>::value)) void lane_pg_fused_kernel(Args g, Knobs kn, const T* __restrict__ recs,
                                                              const uint32_t* __restrict__ order,
                                                              uint32_t* __restrict__ ticket,
                                                              unsigned long long* __restrict__ stats,
                                                              const uint32_t* __restrict__ queue_len) {
    constexpr int RL = LaneRec<T, H>::kLen;
    static_assert(!FAST || Model::kFastStop, "the select-free stop test needs a model with a screen");
    // instances the CD kernel already published sit behind this position of the queue
    const int64_t n_queue = (int64_t)__builtin_nontemporal_load(queue_len);
    // Models with a screen get two builds of this kernel, launched back to back; the coordinate-
    // descent kernel has decided which of them works (stats[2] != 0: some instance failed the
    // screen, the batch takes the exact build) and the other one returns here.
    if constexpr (Model::kFastStop) {
        const bool need_exact = __builtin_nontemporal_load(&stats[2]) != 0ull;
        if (need_exact == FAST) return;
        if (n_queue <= 0) return;   // (nothing queued: no wavefront should go and ask the ticket -- a thousand returning atomics on one address take 35 us)
    }
    // FusedOcc waves per workgroup, each wave an independent solver using its own 64 columns (no
    // barrier anywhere): one 40 KB workgroup per SIMD pair is what the CU is known to co-schedule.
    constexpr int BT = kWave * FusedOcc<T, H>::value;
    constexpr bool REGS = FusedInRegs<T, H>::value;
    // CK: checkpointed w (see FusedCkpt); compact model only -- the general model's linear term
    // routes 2H intermediates through w at every refill
    constexpr bool CK = FusedCkpt<T, H>::value && std::is_same<Model, CompactModel<T>>::value;
    constexpr bool BIG = FusedBig<T, H>::value && !CK;
    constexpr int KV = REGS ? H : (CK ? FusedCkptVRegSteps<T, H>::value : FusedVRegSteps<T, H>::value);   // steps of v in VGPRs
    constexpr int KA = REGS ? 0 : ((BIG || CK) ? H - KV : 0);    // next steps of v in AGPRs
    constexpr int VL = H - KV - KA;                              // the rest of v in LDS
    constexpr bool WA = BIG || CK;                               // w in AGPRs
    // one LDS array, so that a plan without v in LDS costs exactly 2H columns (H = 40: 40 960 bytes
    // per wave, four waves in the CU's 163 840)
    constexpr int MMR = REGS ? 0 : 2 * H, VLR = VL > 0 ? 2 * VL : 0;
    __shared__ T s_all[MMR + VLR > 0 ? MMR + VLR : 1][BT];
    auto s_mm = [&](int q) -> T& { return s_all[q][threadIdx.x]; };
    auto s_v = [&](int q) -> T& { return s_all[MMR + q][threadIdx.x]; };
    // the first KV horizon steps of v stay in VGPRs even when the rest lives in LDS: the inner loop
    // has a few dozen registers to spare, and every step kept saves a ds_read2 and a ds_write2
    // (H = 20 fp64, A/B on one box: KV = 4 / 6 / 8 / 10 / 12 -> +1.7 / +2.0 / +2.1 / +1.5 / -1.0 %)
    T r_mm[REGS ? 2 * H : 1], r_v[2 * KV + 1];   // register-resident copies
    AgprWord a_v[2 * KA + 1], a_w[WA ? (CK ? H : 2 * H) : 1];
    const int lane = threadIdx.x;   // LDS column; ballots below are per wavefront
    auto mm_put = [&](int q, T val) { if constexpr (REGS) r_mm[q] = val; else s_mm(q) = val; };
    auto mm_get = [&](int q) -> T { if constexpr (REGS) return r_mm[q]; else return s_mm(q); };
    auto v_put = [&](int q, T val) {
        if (q < 2 * KV) r_v[q] = val;
        else if (q < 2 * (KV + KA)) agpr_put(a_v[q - 2 * KV], val);
        else s_v(q - 2 * (KV + KA)) = val;
    };
    auto v_get = [&](int q) -> T {
        if (q < 2 * KV) return r_v[q];
        else if (q < 2 * (KV + KA)) return agpr_get<T>(a_v[q - 2 * KV]);
        else return s_v(q - 2 * (KV + KA));
    };
    const T eps = (T)kn.eps;
    // FAST: (u - lo) * 2^600 and (hi - u) * 2^600 as one fma each, exactly zero at the bound and
    // far above any admissible eps off it (>= 4e64; fp32, scale 2^100: >= 7e12).  The bounds of a kFastStop model are the same for the whole batch, so the two
    // addends sit in SGPRs and leave the loop's VGPR budget alone.
    // fp64 goes one step further ("moved" form, 4 slots): the projected-gradient step that is
    // computed anyway says whether a variable is blocked -- v_new = clamp(u - df/lambda) equals u
    // exactly when u sits on a bound with df pushing outward (mpc.h:298-299), or when the step
    // vanishes in rounding, which the screen allows only for |df| < eps.  So
    // min(|df|, |u - v_new| * 2^600) is >= eps exactly where dlib's masked |df| is.
    constexpr bool MOVED = FAST && sizeof(T) == 8;
    constexpr T kHuge = (T)(sizeof(T) == 8 ? 0x1p600 : 0x1p100);
    T nlo_h[2] = {(T)0, (T)0}, hi_h[2] = {(T)0, (T)0};
    // ... and the factor in VGPRs: a VOP3 instruction reads one SGPR operand at most, and with both
    // constants scalar the compiler copies the addend into VGPRs in front of every fma
    T huge = kHuge;
    asm volatile("" : "+v"(huge));
    if constexpr (FAST && !MOVED) {
#pragma unroll
        for (int j = 0; j < I; ++j) {
            nlo_h[j] = wave_uniform(-((T)g.lo[j] * kHuge));
            hi_h[j] = wave_uniform((T)g.hi[j] * kHuge);
        }
    }

    Model m;
    T u[2 * H], w[WA ? 1 : 2 * H];
    // CK keeps the even steps only: step i (even) at slots i, i+1 of a_w, i.e. index (q >> 2) * 2 + (q & 1)
    auto w_put = [&](int q, T val) {
        if constexpr (CK) { if (((q >> 1) & 1) == 0) agpr_put(a_w[(q >> 2) * 2 + (q & 1)], val); }
        else if constexpr (WA) agpr_put(a_w[q], val);
        else w[q] = val;
    };
    auto w_get = [&](int q) -> T {
        if constexpr (CK) return agpr_get<T>(a_w[(q >> 2) * 2 + (q & 1)]);
        else if constexpr (WA) return agpr_get<T>(a_w[q]);
        else return w[q];
    };
    T u0_prev[2] = {(T)0, (T)0};
    T inv_lambda = (T)0, beta = (T)0;
    int64_t k = 0;
    uint32_t iter = 0;
    bool have = false, exhausted = false;
    uint32_t flags = 0;
    uint32_t wave_iters = 0, refills = 0;
#pragma unroll
    for (int q = 0; q < 2 * H; ++q) { u[q] = (T)0; mm_put(q, (T)0); v_put(q, (T)0); w_put(q, (T)0); }

    auto publish = [&](T a0, T a1, uint32_t it) {
        if constexpr (std::is_same<Args, CompactArgs>::value) {
            ((T*)g.front)[k] = a0;
            ((T*)g.rear)[k] = a1;
        } else {
            ((T*)g.u0)[k] = a0;
            if (I == 2) ((T*)g.u0)[g.ld + k] = a1;
        }
        if (g.iters) g.iters[k] = (int32_t)it;
    };

#pragma unroll 1
    while (true) {
        // ---- refill (rare, wave-uniform branch): free lanes pull the next instance of the
        // longest-first queue; instances the CD phase already finished are written out on the spot.
        // (A wave-private pool of 64 tickets per atomic was tried: waves hoard the tail of the
        // queue and the kernel gets slower.)
        // A refill stalls the whole wave for about two iterations (ticket, queue entry, record and
        // inputs are three dependent global round trips, then the linear term is rebuilt), so free
        // lanes wait until RefillBatch<H> of them can be served by one pass; an idle lane costs 1/64
        // of an iteration per iteration, which is far less.
        const unsigned long long want = __ballot(!have && !exhausted);
        if (want != 0ull && (__popcll(want) >= RefillBatch<H>::value || __ballot(have) == 0ull)) {
            ++refills;
            // one atomic per pass, for exactly the tickets it will use (at H = 4 a quarter of a
            // million single-ticket atomics would hit the counter within 0.3 ms)
            const int wl = lane & (kWave - 1);
            uint32_t first_ticket = 0;
            if (wl == __ffsll((long long)want) - 1) first_ticket = atomicAdd(ticket, (uint32_t)__popcll(want));
            first_ticket = (uint32_t)__shfl((int)first_ticket, __ffsll((long long)want) - 1);
            if (!have && !exhausted) {
                const uint32_t t = first_ticket + (uint32_t)__popcll(want & ((1ull << wl) - 1ull));
                if ((int64_t)t >= n_queue) {
                    exhausted = true;
                } else {
                    k = (int64_t)order[t];
                    const T* rec = recs + k * RL;
                    m.load(g, k);   // input loads go out together with the record loads
#pragma unroll
                    for (int q = 0; q < 2 * H; ++q)
                        if ((q & 1) < I) u[q] = rec[q];
                    const T lambda = rec[2 * H];
                    const uint64_t meta = load_meta<T>(rec + 2 * H + 1);
                    iter = (uint32_t)meta;
                    if (meta & kMetaNonFinite) flags |= 0x1u;
                    if (meta & kMetaBadModel) flags |= 0x4u;
                    const bool vinit = (meta & kMetaVInit) != 0;   // mpc.h:330-334, else a fresh v = 0
#pragma unroll
                    for (int q = 0; q < 2 * H; ++q)
                        if ((q & 1) < I) v_put(q, vinit ? u[q] : (T)0);
                    if ((meta & kMetaStopped) || iter >= kn.max_iter) {
                        if (!(meta & kMetaStopped)) flags |= 0x2u;
                        publish(u[0], u[1], iter);
                    } else {
                        auto mm_emit = [&](int q, T val) { mm_put(q, val); };
                        if constexpr (WA && !std::is_same<Model, CompactModel<T>>::value)
                            linear_term_fn<T, I, H>(m, w_put, w_get, mm_emit);
                        else
                            linear_term<T, I, H>(m, w, mm_emit);
                        inv_lambda = (T)1.0 / lambda;                         // mpc.h:342
                        const T sq = tsqrt(lambda);
                        beta = (sq - (T)1) / (sq + (T)1);                     // mpc.h:343
                        have = true;
                    }
                }
            }
        }
        if (__ballot(have) == 0ull) {
            if (__ballot(!exhausted) == 0ull) break;   // every lane saw the end of the queue
            continue;                                  // drew only finished instances: pull again
        }

        // ---- inner loop: iterate until some lane of this wave stops or hits the cap.  Keeping the
        // refill out of this loop leaves u with a single definition on the back-edge, so the
        // controls stay in place instead of being copied every iteration.
        bool stop = false, cap = false;
#pragma unroll 1
        do {
        // ring slot for the last horizon step: fetched now, hidden under the forward pass
        T pm[2][2], pv[2][2];
#pragma unroll
        for (int j = 0; j < I; ++j) {
            pm[(H - 1) & 1][j] = mm_get(2 * (H - 1) + j);
            pv[(H - 1) & 1][j] = v_get(2 * (H - 1) + j);
        }
        // ---- forward pass: M[i] = A*M[i-1] + B*u[i]                       (mpc.h:275-277)
        T m0, m1;
        m.first(m0, m1, &u[0]);
        w_put(0, m0); w_put(1, m1);
#pragma unroll
        for (int i = 1; i < H; ++i) {
            m.fwd(m0, m1, &u[2 * i]);
            w_put(2 * i, m0); w_put(2 * i + 1, m1);
        }
        // ---- backward pass fused with the stop test and the speculative update
        u0_prev[0] = u[0]; u0_prev[1] = u[1];
        constexpr int NA = 4;   // independent max accumulators (max is exact: any split gives dlib's value)
        T acc[NA];
#pragma unroll
        for (int z = 0; z < NA; ++z) acc[z] = (T)0;
        T n0 = m0 * m.Q(0), n1 = m1 * m.Q(1);                                    // mpc.h:278-279 (i = H-1)
        T wc0 = (T)0, wc1 = (T)0;   // CK: the checkpoint M[i-1] read at an odd step i, used again at step i-1
        static_for<H>([&](auto ic) {
            constexpr int i = H - 1 - decltype(ic)::value;
            constexpr int cur = i & 1, nxt = (i - 1) & 1;
            if constexpr (i > 0) {   // prefetch step i-1 while step i computes
                static_for<I>([&](auto jc) {
                    constexpr int j = decltype(jc)::value;
                    pm[nxt][j] = mm_get(2 * (i - 1) + j);
                    pv[nxt][j] = v_get(2 * (i - 1) + j);
                });
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (CK) {
                static_assert(!CK || H % 2 == 0, "checkpoints sit on the even steps");
                if constexpr (i < H - 1) {
                    if constexpr (i % 2 == 1) {
                        // M[i] again from the checkpoint M[i-1] and u[i], which this step has not
                        // updated yet: the forward pass's own operations (mpc.h:277)
                        wc0 = w_get(2 * (i - 1)); wc1 = w_get(2 * (i - 1) + 1);
                        T r0 = wc0, r1 = wc1;
                        m.template fwd<true>(r0, r1, &u[2 * i]);
                        m.bwd(n0, n1, r0, r1);                                           // mpc.h:280-281
                    } else {
                        m.bwd(n0, n1, wc0, wc1);
                    }
                } else {
                    // i = H-1 (odd): M[H-1] is still in m0, m1; fetch the checkpoint for step H-2
                    wc0 = w_get(2 * (i - 1)); wc1 = w_get(2 * (i - 1) + 1);
                }
            } else {
                if constexpr (i < H - 1) m.bwd(n0, n1, w_get(2 * i), w_get(2 * i + 1));   // mpc.h:280-281
            }
            T vn[2], st[2];
            static_for<I>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                constexpr int q = 2 * i + j;
                const T uu = u[q];
                const T dd = (pm[cur][j] + m.btm(j, n0, n1)) + uu * m.R(j);     // mpc.h:283
                if constexpr (MOVED) {
                    vn[j] = clamp3(uu - inv_lambda * dd, m.lo(j), m.hi(j));     // mpc.h:342
                    const T moved = tabs(uu - vn[j]) * huge;
                    acc[(i * I + j) % NA] = tmax(acc[(i * I + j) % NA], tmin(tabs(dd), moved));
                } else if constexpr (FAST) {
                    // mpc.h:298-299 without compares and selects (5 issue slots instead of 8): a
                    // variable at its lower bound may only move up (only df < 0 counts), one at its
                    // upper bound only down.  g_lo / g_hi are 0 at the bound and >= 1e64 off it, so
                    // |clamp(df, -g_hi, g_lo)| is dlib's masked |df| wherever that is below eps
                    // (<= 1e30 by the screen) and some value >= eps wherever it is not; df is never
                    // NaN for a screened instance, and controls never leave [lo, hi] (lo < 0 < hi).
                    const T g_lo = tfma(uu, huge, nlo_h[j]);
                    const T g_hi = tfma(uu, -huge, hi_h[j]);
                    if constexpr (sizeof(T) == 4) {
                        // one med3 per variable, one max3 per horizon step (both variables at once)
                        st[j] = (T)med3_neglo((float)dd, (float)g_hi, (float)g_lo);
                        if constexpr (j == I - 1)
                            acc[i % NA] = (T)max3_abs((float)acc[i % NA], (float)st[0], (float)st[I - 1]);
                    } else {
                        acc[(i * I + j) % NA] = tmax(acc[(i * I + j) % NA], tabs(tmax(tmin(dd, g_lo), -g_hi)));
                    }
                } else {
                    const T up = (uu <= m.lo(j)) ? (T)0 : dd;                   // mpc.h:298-299
                    const T dn = (uu >= m.hi(j)) ? (T)0 : -dd;
                    acc[(i * I + j) % NA] = tmax(acc[(i * I + j) % NA], tmax(up, dn));
                }
                if constexpr (!MOVED) vn[j] = clamp3_fast<FAST>(uu - inv_lambda * dd, m.lo(j), m.hi(j));   // mpc.h:342
                u[q] = clamp3_fast<FAST>(vn[j] + beta * (vn[j] - pv[cur][j]), m.lo(j), m.hi(j));   // mpc.h:343
                asm volatile("" : "+v"(u[q]));   // keep the update in its step (LLVM would sink it)
            });
            static_for<I>([&](auto jc) {   // adjacent stores: one ds_write2st64 per step
                constexpr int j = decltype(jc)::value;
                v_put(2 * i + j, vn[j]);
            });
        });
        T max_df = acc[0];
#pragma unroll
        for (int z = 1; z < NA; ++z) max_df = tmax(max_df, acc[z]);
        ++wave_iters;
        stop = have && (max_df < eps);                                          // mpc.h:310-311
        ++iter;
        cap = have && !stop && iter >= kn.max_iter;                             // mpc.h:271
        // A finished lane publishes right here and the loop goes on; it is left only when a refill
        // pass is due (RefillBatch lanes wait) or no lane has work.  At H = 30 / 40, where the
        // compiler keeps the loop's state in scratch outside the loop, leaving on every stop cost a
        // 300-dword spill and reload per stop (-2 % kernel time, a third of the scratch traffic);
        // -4 % at H = 10.  The fp64 H = 20 kernel keeps round 1's form, which leaves on every stop:
        // there the publish code inside the loop upsets the loop's register allocation (+4.5 %).
        if constexpr (ExitEveryStop<T, H>::value) {
            if (__ballot(stop || cap) != 0ull) break;
        } else if (__ballot(stop || cap) != 0ull) {
            if (stop) {
                publish(u0_prev[0], u0_prev[1], iter - 1);
                have = false;
            }
            if (cap) {
                flags |= 0x2u;
                publish(u[0], u[1], iter);
                have = false;
            }
            const unsigned long long waiting = __ballot(!have && !exhausted);
            if (__popcll(waiting) >= RefillBatch<H>::value || __ballot(have) == 0ull) break;
        }
        } while (true);
        if constexpr (ExitEveryStop<T, H>::value) {
            if (stop) {
                publish(u0_prev[0], u0_prev[1], iter - 1);
                have = false;
            }
            if (cap) {
                flags |= 0x2u;
                publish(u[0], u[1], iter);
                have = false;
            }
        }
    }
    // (at most one atomic per persistent wave either way.  The fp64 H = 40 kernel keeps the bare form: the
    // helper's ballots, though behind the loops, shift its register allocation and cost it 3.4 %, measured;
    // H = 20 and 30 are 1-3 % faster WITH the helper.)
    if constexpr (sizeof(T) == 8 && H == 40) { if (g.flags && flags) atomicOr(g.flags, flags); }
    else raise_flags(g.flags, flags);
    if (stats && (lane & (kWave - 1)) == 0) {
        atomicAdd(&stats[0], (unsigned long long)wave_iters);
        atomicAdd(&stats[1], (unsigned long long)refills);
    }
}

}  // namespace tpc
