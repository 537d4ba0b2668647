// LANE_FMA kernels for the GENERAL model (dlib::mpc<2,I,H> with per-instance A, B, C, Q, R, bounds, x0 and per-step
// targets; mpc.h:51-125, :142-163, :253-347), in the arithmetic of mpc_ubg_model.h.  Same three launches, scratch and
// refill protocol as the compact-form kernels (mpc_ub.h) and LANE (mpc_lane.h), where the design is explained; cold
// starts only (a caller that passes the controller state in or out runs the bit-exact LANE family, whose
// lane_pg_kernel keeps it), horizons 4, 5, 10 and 20 (the general form at N = 30 / 40 stays with LANE).
// Results: dlib's decisions on quantities that differ by rounding (max |du| ~1e-13 at N = 20), iteration counts
// identical on the fixtures and seeded sets; bit-identical to the CPU model tests/model/ub_model.cpp.
#pragma once

#include "mpc_lane.h"
#include "mpc_ubg_model.h"

namespace tpc {

template <typename T, int I> TPC_DEV void ubg_load(ubg::Gen<T, I>& m, const GeneralArgs& g, int64_t k) {
    const T* Ap = (const T*)g.A + k;
    const T* Bp = (const T*)g.B + k;
    m.a00 = Ap[0]; m.a01 = Ap[g.ld]; m.a10 = Ap[2 * g.ld]; m.a11 = Ap[3 * g.ld];
#pragma unroll
    for (int r_ = 0; r_ < 2; ++r_)
#pragma unroll
        for (int j = 0; j < I; ++j) m.b[r_][j] = Bp[(int64_t)(r_ * I + j) * g.ld];
    m.c0 = ((const T*)g.C)[k]; m.c1 = ((const T*)g.C)[g.ld + k];
    m.q0 = ((const T*)g.Q)[k]; m.q1 = ((const T*)g.Q)[g.ld + k];
#pragma unroll
    for (int j = 0; j < I; ++j) {
        m.r[j] = ((const T*)g.R)[(int64_t)j * g.ld + k];
        m.lo[j] = ((const T*)g.lo)[(int64_t)j * g.ld + k];
        m.hi[j] = ((const T*)g.hi)[(int64_t)j * g.ld + k];
    }
    m.x00 = ((const T*)g.x0)[k]; m.x01 = ((const T*)g.x0)[g.ld + k];
}

// ------------------------------------------------------------------------------------------------
// Phase 1: coordinate descent (mpc.h:319-335), 64 instances per wave in lockstep.  1 / Q_diag, the linear
// term MM and a copy of u sit in LDS ([slot][lane], slot = 2 i + j; three arrays: two wavefronts per SIMD at most).
template <typename T, int I, int H>
__global__ __launch_bounds__(64, (CdOcc<T, H>::value > 2 ? 2 : CdOcc<T, H>::value)) void ubg_cd_kernel(GeneralArgs g, Knobs kn, T* __restrict__ recs,
                                                                          uint32_t* __restrict__ keys,
                                                                          uint32_t* __restrict__ key_rank,
                                                                          uint32_t* __restrict__ key_hist,
                                                                          unsigned long long* __restrict__ stats) {
    constexpr int RL = LaneRec<T, H>::kLen;
    __shared__ T s_rqd[2 * H][kWave];
    __shared__ T s_mm[2 * H][kWave];
    __shared__ T s_u[(3 * (2 * H * kWave * (int)sizeof(T)) * 4 * (CdOcc<T, H>::value > 2 ? 2 : CdOcc<T, H>::value) <= 160 * 1024) ? 2 * H : 1][kWave];
    const int lane = threadIdx.x;
    const int64_t k = (int64_t)blockIdx.x * kWave + lane;
    if (k >= g.n) return;

    ubg::Gen<T, I> m;
    ubg_load(m, g, k);
    m.set_scale((T)1);
    const bool nonfinite = m.nonfinite();
    const bool badmodel = m.invalid();
    const T* tg = (const T*)g.targets + k;

    // u in registers; where it costs no residency, a copy in LDS too (the winner is read and written there by index)
    // A third LDS array must not cost residency (4 x waves-per-SIMD wavefronts of a CU share 160 KB): at fp64 N = 20 three
    // arrays of 20 KB leave two wavefronts per CU -- half the SIMDs idle, 0.84 ms against 0.72 -- so there the copy is
    // dropped and the winner's u is tracked by select, its update written by a select chain.
    constexpr int kCdWaves = CdOcc<T, H>::value > 2 ? 2 : CdOcc<T, H>::value;
    constexpr bool MIRROR = 3 * (2 * H * kWave * (int)sizeof(T)) * 4 * kCdWaves <= 160 * 1024;
    T u[2 * H], w[2 * H];
#pragma unroll
    for (int q = 0; q < 2 * H; ++q) {
        u[q] = (T)0;
        if constexpr (MIRROR) s_u[q][lane] = (T)0;
    }
    const T lambda = ubg::ctor_lambda_qdiag<T, I, H>(m, [&](int i, int j, T val) {
        s_rqd[2 * i + j][lane] = val != (T)0 ? (T)1 / val : (T)0;   // mpc.h:322: a zero Q_diag never updates
    });
    T mm_max = (T)0;
    bool mm_nan = false;
    ubg::linear_term<T, I, H>(
        m, [&](int i, int s) { return tg[(int64_t)(2 * i + s) * g.ld]; }, [&](int q, T val) { w[q] = val; },
        [&](int q) { return w[q]; },
        [&](int i, int j, T val) { s_mm[2 * i + j][lane] = val; mm_max = tmax(mm_max, tabs(val)); mm_nan = mm_nan || val != val; });
    if (mm_nan) mm_max = (T)__builtin_inf();
    const T eps = (T)kn.eps;
    const unsigned long long failing = __ballot(!ubg::fast_stop_ok(m, mm_max, eps, lambda, H));
    if (failing != 0ull && lane == __ffsll((long long)failing) - 1) atomicOr(&stats[2], 1ull);
    const uint32_t cd_iters = kn.smo_iters < kn.max_iter ? kn.smo_iters : kn.max_iter;
    uint32_t iter = 0;
    bool stopped = badmodel;   // (dlib propagates non-finite values through this arithmetic itself: flagged, not screened)
    bool vinit = false;
    constexpr T kHuge = (T)(sizeof(T) == 8 ? 0x1p600 : 0x1p100);
    T huge = kHuge;
    asm volatile("" : "+v"(huge));
    T lo_h[I], hi_h[I];
#pragma unroll
    for (int j = 0; j < I; ++j) { lo_h[j] = -(m.lo[j] * huge); hi_h[j] = m.hi[j] * huge; }

    // One iteration, in the two builds of ub_cd_kernel (mpc_ub.h): `fast` where every lane of the wavefront passed the
    // screen -- dlib's mask as arithmetic on the gaps (u - lower) 2^600 and (upper - u) 2^600, one signed running
    // arg-max -- else dlib's compares and selects.  Same decisions, same u.
    auto iteration = [&](auto fast_tag, uint32_t it) {
        constexpr bool F = decltype(fast_tag)::value;
        T m0, m1;
        m.first(m0, m1, &u[0]);
        w[0] = m0; w[1] = m1;
#pragma unroll
        for (int i = 1; i < H; ++i) {
            m.fwd(m0, m1, &u[2 * i]);
            w[2 * i] = m0; w[2 * i + 1] = m1;
        }
        T n0, n1;
        m.bwd_last(n0, n1, m0, m1);
        // arg-max fused into the backward sweep: descending with '>=' picks what dlib's ascending strict '>' picks
        T max_df = (T)0, best_mm = (T)0, best_u = (T)0;
        int best = 0, best_sign = 0;
#pragma unroll
        for (int i = H - 1; i >= 0; --i) {
            if (i < H - 1) m.bwd(n0, n1, w[2 * i], w[2 * i + 1]);
            const T ui[2] = {u[2 * i], u[2 * i + 1]};
#pragma unroll
            for (int j = I - 1; j >= 0; --j) {
                const int q = 2 * i + j;
                const T uu = ui[j];
                const T dd = m.df(j, n0, n1, uu, s_mm[q][lane]);
                if constexpr (F) {
                    const T g_lo = ub::fma_(uu, huge, lo_h[j]), g_hi = ub::fma_(-huge, uu, hi_h[j]);
                    const T mm = tmax(tmin(dd, g_lo), -g_hi);         // df where dlib counts it (mpc.h:298-299), else 0
                    const bool better = tabs(mm) >= tabs(best_mm);   // (zeros may pass one another: all of them mean 'none')
                    best_mm = better ? mm : best_mm;
                    best = better ? q : best;
                    if constexpr (!MIRROR) best_u = better ? uu : best_u;
                } else {
                    const T up = (uu <= m.lo[j]) ? (T)0 : dd;
                    const T dn = (uu >= m.hi[j]) ? (T)0 : -dd;
                    const T mag = tmax(up, dn);
                    const bool better = mag >= max_df && mag > (T)0;
                    max_df = tmax(max_df, mag);
                    best = better ? q : best;
                    best_sign = better ? sign_word(dd) : best_sign;
                    if constexpr (!MIRROR) best_u = better ? uu : best_u;
                }
            }
        }
        if constexpr (F) max_df = tabs(best_mm);
        stopped = stopped || max_df < eps;                      // mpc.h:310-311
        const bool act = !stopped;
        T best_df;
        if constexpr (F) best_df = best_mm; else best_df = with_sign(max_df, best_sign);
        if constexpr (MIRROR) best_u = s_u[MIRROR ? best : 0][lane];
        const T rq = s_rqd[best][lane];
        const bool upd = act && rq != (T)0;                     // mpc.h:322 (`continue` still counts)
        // (bounds picked by select: a run-time index would put the model in scratch)
        const bool second = I == 2 && (best & 1);
        const T blo = second ? m.lo[I - 1] : m.lo[0], bhi = second ? m.hi[I - 1] : m.hi[0];
        const T nu = tmax(tmin(ub::fma_(-rq, best_df, best_u), bhi), blo);   // mpc.h:325-326
        if constexpr (MIRROR) { if (upd) s_u[MIRROR ? best : 0][lane] = nu; }
        const int sel = upd ? best : -1;
#pragma unroll
        for (int q = 0; q < 2 * H; ++q)
            if ((q & 1) < I) u[q] = (q == sel) ? nu : u[q];
        vinit = upd ? (it + 1 == kn.smo_iters) : vinit;         // mpc.h:330-334
        iter += act ? 1u : 0u;
    };
    if (failing == 0ull) {
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
    const bool finished = stopped || iter >= kn.max_iter;
    T lambda_floor = m.r[0];
    if (I == 2) lambda_floor = lambda_floor + m.r[I - 1];
    lambda_floor = lambda_floor * (T)H;
    const bool uninformative = lambda < (T)1.5 * lambda_floor;
    const float lf = g.work_hint ? (float)(g.work_hint[k] > 0 ? g.work_hint[k] : 1) : (float)lambda;
    const uint32_t spread = (uint32_t)k & 127u;
    uint32_t key = __float_as_uint(lf);
    if (finished) key = spread << 16;
    else if (!g.work_hint && uninformative) key = (0x7f00u + spread) << 16;
    else if (!(lf > 0.0f) || key < 0x00800000u) key = 0x00800000u;
    else if (key >= 0x7f000000u) key = 0x7effffffu;
    uint32_t f = 0;
    if (finished) {
#pragma unroll
        for (int j = 0; j < I; ++j) ((T*)g.u0)[(int64_t)j * g.ld + k] = u[j];
        if (g.iters) g.iters[k] = (int32_t)iter;
        if (nonfinite) f |= 0x1u;
        if (badmodel) f |= 0x4u;
        if (!stopped) f |= 0x2u;
    }
    raise_flags(g.flags, f);
    keys[k] = key;
    key_rank[k] = atomicAdd(&key_hist[key >> 16], 1u);
}

// ------------------------------------------------------------------------------------------------
// Phase 2: the fused projected-gradient kernel (structure: lane_pg_fused_kernel / ub_pg_kernel).  u and the
// forward pass in VGPRs; g MM and v in VGPRs up to N = 10 (fp32: always), at N = 20 fp64 MM in LDS and v in VGPRs
// for the first KV steps, in LDS beyond, both fetched one step ahead.
template <typename T, int H> struct UbgPlan {
    static constexpr bool D = sizeof(T) == 8;
    static constexpr bool regs = !D || H <= 10;
    static constexpr int occ = (!D && H == 10) ? 2 : 1;
    static constexpr int kv = regs ? H : 6;
};

template <typename T, int I, int H, bool FAST>
__global__ __launch_bounds__((64 * UbgPlan<T, H>::occ), (UbgPlan<T, H>::occ)) void ubg_pg_kernel(
    GeneralArgs g, Knobs kn, const T* __restrict__ recs, const uint32_t* __restrict__ order,
    uint32_t* __restrict__ ticket, unsigned long long* __restrict__ stats, const uint32_t* __restrict__ queue_len) {
    using P = UbgPlan<T, H>;
    constexpr int RL = LaneRec<T, H>::kLen;
    const int64_t n_queue = (int64_t)__builtin_nontemporal_load(queue_len);
    {
        const bool need_exact = __builtin_nontemporal_load(&stats[2]) != 0ull;
        if (need_exact == FAST) return;
        if (n_queue <= 0) return;   // (nothing queued: no wavefront should go and ask the ticket -- a thousand returning atomics on one address take 35 us)
    }
    constexpr int BT = kWave * P::occ;
    constexpr bool REGS = P::regs;
    constexpr int KV = P::kv, VL = H - KV;
    constexpr int MMR = REGS ? 0 : 2 * H, VLR = VL > 0 ? 2 * VL : 0;
    __shared__ T s_all[MMR + VLR > 0 ? MMR + VLR : 1][BT];
    T r_mm[REGS ? 2 * H : 1], r_v[2 * KV + 1];
    const int lane = threadIdx.x;
    auto mm_put = [&](int q, T val) { if constexpr (REGS) r_mm[q] = val; else s_all[q][threadIdx.x] = val; };
    auto mm_get = [&](int q) -> T { if constexpr (REGS) return r_mm[q]; else return s_all[q][threadIdx.x]; };
    auto v_put = [&](int q, T val) { if (q < 2 * KV) r_v[q] = val; else s_all[MMR + q - 2 * KV][threadIdx.x] = val; };
    auto v_get = [&](int q) -> T { if (q < 2 * KV) return r_v[q]; else return s_all[MMR + q - 2 * KV][threadIdx.x]; };

    constexpr T gs = ubg::GradScale<T>::g;
    const T geps = gs * (T)kn.eps;
    constexpr bool MOVED = FAST && sizeof(T) == 8;
    T huge = (T)0x1p100;
    asm volatile("" : "+v"(huge));

    ubg::Gen<T, I> m;
    T u[2 * H], w[2 * H];
    T u0_prev[2] = {(T)0, (T)0};
    T il = (T)0, beta = (T)0;
    int64_t k = 0;
    uint32_t iter = 0;
    bool have = false, exhausted = false;
    uint32_t flags = 0;
    uint32_t wave_iters = 0, refills = 0;
#pragma unroll
    for (int q = 0; q < 2 * H; ++q) { u[q] = (T)0; w[q] = (T)0; mm_put(q, (T)0); v_put(q, (T)0); }

    auto publish = [&](T a0, T a1, uint32_t it) {
        ((T*)g.u0)[k] = a0;
        if (I == 2) ((T*)g.u0)[g.ld + k] = a1;
        if (g.iters) g.iters[k] = (int32_t)it;
    };

#pragma unroll 1
    while (true) {
        const unsigned long long want = __ballot(!have && !exhausted);
        if (want != 0ull && (__popcll(want) >= RefillBatch<H>::value || __ballot(have) == 0ull)) {
            ++refills;
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
                    ubg_load(m, g, k);
                    m.set_scale(gs);
#pragma unroll
                    for (int q = 0; q < 2 * H; ++q)
                        if ((q & 1) < I) u[q] = rec[q];
                    const T lambda = rec[2 * H];
                    const uint64_t meta = load_meta<T>(rec + 2 * H + 1);
                    iter = (uint32_t)meta;
                    if (meta & kMetaNonFinite) flags |= 0x1u;
                    if (meta & kMetaBadModel) flags |= 0x4u;
                    const bool vinit = (meta & kMetaVInit) != 0;
#pragma unroll
                    for (int q = 0; q < 2 * H; ++q)
                        if ((q & 1) < I) v_put(q, vinit ? u[q] : (T)0);
                    if ((meta & kMetaStopped) || iter >= kn.max_iter) {
                        if (!(meta & kMetaStopped)) flags |= 0x2u;
                        publish(u[0], u[1], iter);
                    } else {
                        const T* tg = (const T*)g.targets + k;
                        ubg::linear_term<T, I, H>(
                            m, [&](int i, int s) { return tg[(int64_t)(2 * i + s) * g.ld]; }, [&](int q, T val) { w[q] = val; },
                            [&](int q) { return w[q]; }, [&](int i, int j, T val) { mm_put(2 * i + j, val); });
                        il = ((T)1 / lambda) * ubg::GradScale<T>::inv_g;          // mpc.h:342
                        const T sq = tsqrt(lambda);
                        beta = (sq - (T)1) / (sq + (T)1);                        // mpc.h:343
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
        T pm[2][2], pv[2][2];
        if constexpr (!REGS) {
#pragma unroll
            for (int j = 0; j < I; ++j) {
                pm[(H - 1) & 1][j] = mm_get(2 * (H - 1) + j);
                if constexpr (H - 1 >= KV) pv[(H - 1) & 1][j] = v_get(2 * (H - 1) + j);
            }
        }
        // ---- forward pass (mpc.h:275-277)
        T m0, m1;
        m.first(m0, m1, &u[0]);
        w[0] = m0; w[1] = m1;
#pragma unroll
        for (int i = 1; i < H; ++i) {
            m.fwd(m0, m1, &u[2 * i]);
            w[2 * i] = m0; w[2 * i + 1] = m1;
        }
        // ---- backward pass fused with the stop test and the speculative update
        u0_prev[0] = u[0]; u0_prev[1] = u[1];
        constexpr int NA = 4;
        T acc[NA];
#pragma unroll
        for (int z = 0; z < NA; ++z) acc[z] = (T)0;
        T n0, n1;
        m.bwd_last(n0, n1, m0, m1);
        static_for<H>([&](auto ic) {
            constexpr int i = H - 1 - decltype(ic)::value;
            constexpr int cur = i & 1, nxt = (i - 1) & 1;
            if constexpr (!REGS && i > 0) {
                static_for<I>([&](auto jc) {
                    constexpr int j = decltype(jc)::value;
                    pm[nxt][j] = mm_get(2 * (i - 1) + j);
                    if constexpr (i - 1 >= KV) pv[nxt][j] = v_get(2 * (i - 1) + j);
                });
                __builtin_amdgcn_sched_barrier(0);   // pins the prefetch at the top of its step (14 steps of v + 20 of MM in LDS)
            }
            if constexpr (i < H - 1) m.bwd(n0, n1, w[2 * i], w[2 * i + 1]);               // mpc.h:280-281
            T vn[2], st[2];
            static_for<I>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                constexpr int q = 2 * i + j;
                const T uu = u[q];
                T gmm;
                if constexpr (REGS) gmm = r_mm[q]; else gmm = pm[cur][j];
                const T dd = m.df(j, n0, n1, uu, gmm);                                   // mpc.h:283
                vn[j] = m.template project<FAST>(ub::fma_(-il, dd, uu), j);              // mpc.h:342
                if constexpr (MOVED) {
                    acc[(i * I + j) % NA] = tmax(acc[(i * I + j) % NA], tmin(tabs(dd), tabs(uu - vn[j])));
                } else if constexpr (FAST) {
                    const T g_lo = ub::fma_(uu, huge, -(m.lo[j] * huge));
                    const T g_hi = ub::fma_(-huge, uu, m.hi[j] * huge);
                    st[j] = (T)med3_neglo((float)dd, (float)g_hi, (float)g_lo);
                    if constexpr (j == I - 1) acc[i % NA] = (T)max3_abs((float)acc[i % NA], (float)st[0], (float)st[I - 1]);
                } else {
                    const T up = (uu <= m.lo[j]) ? (T)0 : dd;                             // mpc.h:298-299
                    const T dn = (uu >= m.hi[j]) ? (T)0 : -dd;
                    acc[(i * I + j) % NA] = tmax(acc[(i * I + j) % NA], tmax(up, dn));
                }
            });
            static_for<I>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                constexpr int q = 2 * i + j;
                T vold;
                if constexpr (i >= KV) vold = pv[cur][j]; else vold = r_v[q];
                u[q] = m.template project<FAST>(ub::fma_(beta, vn[j] - vold, vn[j]), j);  // mpc.h:343 (difference form)
                if constexpr (i < KV) r_v[q] = vn[j];
            });
            if constexpr (i >= KV) {
                static_for<I>([&](auto jc) {
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
    }
    raise_flags(g.flags, flags);
    if (stats && (lane & (kWave - 1)) == 0) {
        atomicAdd(&stats[0], (unsigned long long)wave_iters);
        atomicAdd(&stats[1], (unsigned long long)refills);
    }
}

}  // namespace tpc
