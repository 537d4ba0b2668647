// GROUP kernel for the GENERAL model: G lanes per instance (mpc_group.h is the compact form's), in the arithmetic of
// mpc_ubg_model.h -- dlib::mpc<2,I,H> with per-instance A, B, C, Q, R, bounds, x0 and per-step targets (reference:
// dlib_files/dlib/control/mpc.h:51-125, :142-163, :253-347): the accelerated projected-gradient phase (mpc.h:336-345)
// behind a coordinate-descent kernel that leaves the usual records (ubg_cd_kernel; lane_cd_kernel where the caller
// passes the controller state or the horizon has no ubg kernel).
//
// What differs from the compact form.  A is a full 2 x 2 matrix per instance, so joining the chunks of a recurrence
// needs its powers: the chunk of L steps maps a carried state s to A^L s + (what the chunk adds), and combining with the
// chunk d lanes away is  s += A^(L d) s'  -- a 2 x 2 matrix-vector product (4 fused multiply-adds on two DPP-moved
// values) instead of the compact form's `z += z' + (d L a) y'`.  A^L, A^2L, A^4L are formed once per instance (with the
// 0 / 1 weight of the scan step folded in); the backward recurrence uses their transposes.  The correction of the L
// local values by the carried state is the carried state pushed through A step by step (4 fused multiply-adds and two
// additions per step).  The linear term (mpc.h:258-266: per-step targets, a free x0, C) is computed by the same
// machinery once per instance -- the chunk's free response, a scan, the chunk's part of the backward accumulation, a
// scan -- and kept, scaled, in registers (2 L values).
//
// Controller state.  STATE = true serves callers that pass controls_inout / v_inout (warm-start chains,
// tpc_mpc_rollout): dlib's v comes from the caller where the coordinate-descent phase did not set it (mpc.h:330-334),
// and the whole solved sequence and v go back, which takes a copy of the chunk's controls per iteration (the update is
// speculative; a group that stops publishes what it had before it).  Only L values per lane: what made the one-lane
// families fall back to the unfused bit-exact kernel costs this layout ten registers.
//
// Results: the LANE_FMA statement -- dlib's decisions on quantities that differ from dlib's by rounding, <= 1e-9 and
// identical iteration counts against the oracle and the real-dlib fixtures (tests/test_group_gpu.py).  Screened stop
// test only; a batch the screen refuses runs the exact build of the one-lane family on the same records.
#pragma once

#include "mpc_group.h"
#include "mpc_ubg.h"

namespace tpc {

template <typename T, int I, int H, int G, bool STATE>
__global__ __launch_bounds__(64, 1) void groupg_pg_kernel(GeneralArgs g, Knobs kn, const T* __restrict__ recs,
                                                          const uint32_t* __restrict__ order, uint32_t* __restrict__ ticket,
                                                          unsigned long long* __restrict__ stats,
                                                          const uint32_t* __restrict__ queue_len) {
    using P = GroupPlan<T, H, G>;
    constexpr int L = P::L, RL = LaneRec<T, H>::kLen, NV = I * L;
    constexpr bool D64 = sizeof(T) == 8;
    // (STATE: the coordinate-descent kernel publishes nothing itself -- the caller wants controls and v of every instance --
    // so the whole queue is this kernel's, finished instances included)
    const int64_t n_queue = STATE ? g.n : (int64_t)__builtin_nontemporal_load(queue_len);
    if (__builtin_nontemporal_load(&stats[2]) != 0ull) return;   // the screen refused the batch: the exact build's
    if (n_queue <= 0) return;

    const int lane = threadIdx.x;
    const int p = lane & (G - 1);
    const int gbase = lane & ~(G - 1);
    constexpr T gs = ubg::GradScale<T>::g;
    const T geps = gs * (T)kn.eps;
    T huge = (T)0x1p100;
    asm volatile("" : "+v"(huge));

    constexpr int KS = P::steps, KA = KS > 0 ? KS : 1;
    constexpr int DL0 = P::dl0, ND = L - DL0 > 0 ? L - DL0 : 1;
    const T wf0 = p >= 1 ? (T)1 : (T)0, wb0 = p + 1 < G ? (T)1 : (T)0;
    T live[ND];
#pragma unroll
    for (int l = DL0; l < L; ++l) live[l - DL0] = p * L + l < H ? (T)1 : (T)0;

    ubg::Gen<T, I> m;
    m.a00 = m.a01 = m.a10 = m.a11 = m.c0 = m.c1 = m.q0 = m.q1 = m.gq0 = m.gq1 = m.x00 = m.x01 = (T)0;
#pragma unroll
    for (int j = 0; j < I; ++j) { m.b[0][j] = m.b[1][j] = m.r[j] = m.gr[j] = m.lo[j] = m.hi[j] = (T)0; }
    // per instance: weighted powers of A (row-major) for the scan steps; forward uses pf[s] = w A^(L 2^s), backward the
    // transpose with its own weight
    T pf[KA][4], pb[KA][4];
#pragma unroll
    for (int s = 0; s < KA; ++s)
#pragma unroll
        for (int e = 0; e < 4; ++e) pf[s][e] = pb[s][e] = (T)0;
    T gmm[NV];              // g MM of the chunk (mpc.h:258-266)
    T x[NV], v[NV], v2[NV], xs[STATE ? NV : 1];
    T x0_prev[2] = {(T)0, (T)0};
    T il = (T)0, beta = (T)0;
    int64_t k = 0;
    uint32_t iter = 0;
    bool have = false, exhausted = false;
    uint32_t flags = 0;
    uint32_t wave_iters = 0, refills = 0;
#pragma unroll
    for (int q = 0; q < NV; ++q) { x[q] = v[q] = v2[q] = gmm[q] = (T)0; if constexpr (STATE) xs[q] = (T)0; }

    // ---- the two scans (exclusive, over the lanes of the group) of a carried 2-vector
    auto scan_fwd = [&](T s0, T s1, T& e0, T& e1) {
        e0 = wf0 * group_mov<GroupDpp<G, 1, false>::ctrl>(s0);
        e1 = wf0 * group_mov<GroupDpp<G, 1, false>::ctrl>(s1);
        static_for<KS>([&](auto sc) {
            constexpr int s = decltype(sc)::value;
            constexpr int ctrl = GroupDpp<G, (1 << s), false>::ctrl;
            const T o0 = group_mov<ctrl>(e0), o1 = group_mov<ctrl>(e1);
            e0 = ub::fma_(pf[s][0], o0, ub::fma_(pf[s][1], o1, e0));
            e1 = ub::fma_(pf[s][2], o0, ub::fma_(pf[s][3], o1, e1));
        });
    };
    auto scan_bwd = [&](T s0, T s1, T& f0, T& f1) {
        f0 = wb0 * group_mov<GroupDpp<G, 1, true>::ctrl>(s0);
        f1 = wb0 * group_mov<GroupDpp<G, 1, true>::ctrl>(s1);
        static_for<KS>([&](auto sc) {
            constexpr int s = decltype(sc)::value;
            constexpr int ctrl = GroupDpp<G, (1 << s), true>::ctrl;
            const T o0 = group_mov<ctrl>(f0), o1 = group_mov<ctrl>(f1);
            f0 = ub::fma_(pb[s][0], o0, ub::fma_(pb[s][1], o1, f0));
            f1 = ub::fma_(pb[s][2], o0, ub::fma_(pb[s][3], o1, f1));
        });
    };
    // stage weights of a local step (0 for a dummy step of a padded chunk)
    auto gq0_at = [&](int l) -> T { return l >= DL0 ? m.gq0 * live[l - DL0] : m.gq0; };
    auto gq1_at = [&](int l) -> T { return l >= DL0 ? m.gq1 * live[l - DL0] : m.gq1; };

    auto publish = [&](const T* xa, const T* va, uint32_t it) {
        if (p == 0) {
#pragma unroll
            for (int j = 0; j < I; ++j) ((T*)g.u0)[(int64_t)j * g.ld + k] = xa[j];
            if (g.iters) g.iters[k] = (int32_t)it;
        }
        if constexpr (STATE) {
#pragma unroll
            for (int l = 0; l < L; ++l)
#pragma unroll
                for (int j = 0; j < I; ++j) {
                    const int i = p * L + l;
                    if (i < H) {
                        if (g.controls) ((T*)g.controls)[(int64_t)(i * I + j) * g.ld + k] = xa[l * I + j];
                        if (g.v) ((T*)g.v)[(int64_t)(i * I + j) * g.ld + k] = va[l * I + j];
                    }
                }
        }
    };

#pragma unroll 1
    while (true) {
        const unsigned long long want = ballot_b(!have && !exhausted);
        if (want != 0ull && (__popcll(want) >= GroupRefillBatch<G>::value * G || ballot_b(have) == 0ull)) {
            ++refills;
            const uint32_t cnt = (uint32_t)__popcll(want) / G;
            const uint32_t rank = (uint32_t)__popcll(want & ((1ull << gbase) - 1ull)) / G;
            const int leader = __ffsll((long long)want) - 1;
            uint32_t first_ticket = 0;
            if (lane == leader) first_ticket = atomicAdd(ticket, cnt);
            first_ticket = (uint32_t)__shfl((int)first_ticket, leader);
            const bool mine = !have && !exhausted;
            bool fresh = false;
            if (mine) {
                const uint32_t t = first_ticket + rank;
                if ((int64_t)t >= n_queue) {
                    exhausted = true;
                } else {
                    k = (int64_t)order[t];
                    const T* rec = recs + k * RL;
                    ubg_load(m, g, k);
                    m.set_scale(gs);
#pragma unroll
                    for (int l = 0; l < L; ++l)
#pragma unroll
                        for (int j = 0; j < I; ++j) x[l * I + j] = p * L + l < H ? rec[2 * (p * L + l) + j] : (T)0;
                    const T lambda = rec[2 * H];
                    const uint64_t meta = load_meta<T>(rec + 2 * H + 1);
                    iter = (uint32_t)meta;
                    if (meta & kMetaNonFinite) flags |= 0x1u;
                    if (meta & kMetaBadModel) flags |= 0x4u;
                    const bool vinit = (meta & kMetaVInit) != 0;   // mpc.h:330-334, else the controller's own v
#pragma unroll
                    for (int l = 0; l < L; ++l)
#pragma unroll
                        for (int j = 0; j < I; ++j) {
                            T vin = (T)0;
                            if constexpr (STATE) {
                                const int i = p * L + l;
                                if (g.v && i < H) vin = ((const T*)g.v)[(int64_t)(i * I + j) * g.ld + k];
                            }
                            v[l * I + j] = vinit ? x[l * I + j] : vin;
                        }
                    if ((meta & kMetaStopped) || iter >= kn.max_iter) {
                        // (the coordinate-descent kernel publishes these itself unless the caller wants the state back)
                        if (!(meta & kMetaStopped)) flags |= 0x2u;
                        publish(x, v, iter);
                    } else {
                        il = ((T)1 / lambda) * ubg::GradScale<T>::inv_g;          // mpc.h:342
                        const T sq = tsqrt(lambda);
                        beta = (sq - (T)1) / (sq + (T)1);                        // mpc.h:343
                        // A^L by L - 1 products, then squarings; weights folded in
                        T q00 = m.a00, q01 = m.a01, q10 = m.a10, q11 = m.a11;
#pragma unroll
                        for (int l = 1; l < L; ++l) {
                            const T n00 = ub::fma_(q00, m.a00, q01 * m.a10), n01 = ub::fma_(q00, m.a01, q01 * m.a11);
                            const T n10 = ub::fma_(q10, m.a00, q11 * m.a10), n11 = ub::fma_(q10, m.a01, q11 * m.a11);
                            q00 = n00; q01 = n01; q10 = n10; q11 = n11;
                        }
#pragma unroll
                        for (int s = 0; s < KS; ++s) {
                            const T wfs = p >= (1 << s) ? (T)1 : (T)0, wbs = p + (1 << s) < G ? (T)1 : (T)0;
                            pf[s][0] = wfs * q00; pf[s][1] = wfs * q01; pf[s][2] = wfs * q10; pf[s][3] = wfs * q11;
                            pb[s][0] = wbs * q00; pb[s][1] = wbs * q10; pb[s][2] = wbs * q01; pb[s][3] = wbs * q11;   // transpose
                            const T n00 = ub::fma_(q00, q00, q01 * q10), n01 = ub::fma_(q00, q01, q01 * q11);
                            const T n10 = ub::fma_(q10, q00, q11 * q10), n11 = ub::fma_(q10, q01, q11 * q11);
                            q00 = n00; q01 = n01; q10 = n10; q11 = n11;
                        }
                        have = true;
                        fresh = true;
                    }
                }
            }
            // ---- the linear term of the fresh instances (mpc.h:258-266), every lane of the wavefront in step (the scans
            //      are wavefront-wide; lanes without a fresh instance compute on what they hold and keep nothing)
            if (ballot_b(fresh) != 0ull) {
                const T* tg = (const T*)g.targets + k;
                // the free response of the chunk: M = A M + C, from x0 in the first chunk
                T fz[L], fy[L];
                T m0 = p == 0 ? m.x00 : (T)0, m1 = p == 0 ? m.x01 : (T)0;
                const T cc0 = m.c0, cc1 = m.c1;
#pragma unroll
                for (int l = 0; l < L; ++l) {
                    const T n0 = ub::fma_(m.a00, m0, ub::fma_(m.a01, m1, cc0));
                    const T n1 = ub::fma_(m.a10, m0, ub::fma_(m.a11, m1, cc1));
                    m0 = n0; m1 = n1;
                    fz[l] = m0; fy[l] = m1;
                }
                T e0, e1;
                scan_fwd(m0, m1, e0, e1);
                // Q .* (M - target), accumulated backwards with trans(A)
                T rz[L], ry[L];
                T n0 = (T)0, n1 = (T)0;
                T c0 = e0, c1 = e1;
#pragma unroll
                for (int l = 0; l < L; ++l) {   // the carried state pushed through the chunk: A^(l+1) e
                    const T d0 = ub::fma_(m.a00, c0, m.a01 * c1), d1 = ub::fma_(m.a10, c0, m.a11 * c1);
                    c0 = d0; c1 = d1;
                    const int i = p * L + l;
                    T t0 = (T)0, t1 = (T)0;
                    if (fresh && i < H) { t0 = tg[(int64_t)(2 * i) * g.ld]; t1 = tg[(int64_t)(2 * i + 1) * g.ld]; }
                    rz[l] = gq0_at(l) * ((fz[l] + c0) - t0);
                    ry[l] = gq1_at(l) * ((fy[l] + c1) - t1);
                }
                T bz[L], by[L];
#pragma unroll
                for (int l = L - 1; l >= 0; --l) {
                    const T t0 = rz[l] + ub::fma_(m.a00, n0, m.a10 * n1);
                    const T t1 = ry[l] + ub::fma_(m.a01, n0, m.a11 * n1);
                    n0 = t0; n1 = t1;
                    bz[l] = n0; by[l] = n1;
                }
                T f0, f1;
                scan_bwd(n0, n1, f0, f1);
                T d0 = f0, d1 = f1;
#pragma unroll
                for (int l = L - 1; l >= 0; --l) {   // trans(A)^(L-l) f
                    const T h0 = ub::fma_(m.a00, d0, m.a10 * d1), h1 = ub::fma_(m.a01, d0, m.a11 * d1);
                    d0 = h0; d1 = h1;
                    const T N0 = bz[l] + d0, N1 = by[l] + d1;
#pragma unroll
                    for (int j = 0; j < I; ++j) {
                        const T val = ub::fma_(m.b[0][j], N0, m.b[1][j] * N1);
                        if (fresh) gmm[l * I + j] = val;
                    }
                }
            }
        }
        if (ballot_b(have) == 0ull) {
            if (ballot_b(!exhausted) == 0ull) break;
            continue;
        }

        auto iteration = [&](T (&vi)[NV], T (&vo)[NV]) -> bool {
            // ---- forward recurrence of the chunk from a zero state (mpc.h:275-277)
            T wz[L], wy[L];
            T m0 = (T)0, m1 = (T)0;
#pragma unroll
            for (int l = 0; l < L; ++l) {
                m.fwd(m0, m1, &x[l * I]);
                wz[l] = m0; wy[l] = m1;
            }
            T e0, e1;
            scan_fwd(m0, m1, e0, e1);
            // ---- backward recurrence on the corrected states, from a zero costate (mpc.h:278-281)
            T cz[L], cy[L];
            {
                T c0 = e0, c1 = e1;
#pragma unroll
                for (int l = 0; l < L; ++l) {
                    const T d0 = ub::fma_(m.a00, c0, m.a01 * c1), d1 = ub::fma_(m.a10, c0, m.a11 * c1);
                    c0 = d0; c1 = d1;
                    cz[l] = wz[l] + c0; cy[l] = wy[l] + c1;
                }
            }
            if constexpr (STATE) {
#pragma unroll
                for (int q = 0; q < NV; ++q) xs[q] = x[q];
            }
            x0_prev[0] = x[0];
            if (I == 2) x0_prev[1] = x[I - 1];
            T nl0[L], nl1[L];
            T n0 = (T)0, n1 = (T)0;
#pragma unroll
            for (int l = L - 1; l >= 0; --l) {
                const T t0 = ub::fma_(gq0_at(l), cz[l], ub::fma_(m.a00, n0, m.a10 * n1));
                const T t1 = ub::fma_(gq1_at(l), cy[l], ub::fma_(m.a01, n0, m.a11 * n1));
                n0 = t0; n1 = t1;
                nl0[l] = n0; nl1[l] = n1;
            }
            T f0, f1;
            scan_bwd(n0, n1, f0, f1);
            // ---- gradient (mpc.h:283), stop test (mpc.h:289-311), speculative update (mpc.h:342-343)
            T acc0 = (T)0, acc1 = (T)0;
            T d0 = f0, d1 = f1;
#pragma unroll
            for (int l = L - 1; l >= 0; --l) {
                const T h0 = ub::fma_(m.a00, d0, m.a10 * d1), h1 = ub::fma_(m.a01, d0, m.a11 * d1);
                d0 = h0; d1 = h1;
                const T N0 = nl0[l] + d0, N1 = nl1[l] + d1;
#pragma unroll
                for (int j = 0; j < I; ++j) {
                    const int q = l * I + j;
                    const T uu = x[q];
                    T dd = m.df(j, N0, N1, uu, gmm[q]);
                    if (l >= DL0) dd = dd * live[l - DL0];
                    const T vn = m.template project<true>(ub::fma_(-il, dd, uu), j);            // mpc.h:342
                    T& acc = j == 0 ? acc0 : acc1;
                    if constexpr (STATE) {
                        // a caller's warm start may lie OUTSIDE the box (the coordinate-descent phase clamps only what
                        // it updates), where "blocked" cannot be read off the projected step: dlib's own mask (mpc.h:298-299)
                        const T up = (uu <= m.lo[j]) ? (T)0 : dd;
                        const T dn = (uu >= m.hi[j]) ? (T)0 : -dd;
                        acc = tmax(acc, tmax(up, dn));
                    } else if constexpr (D64) {
                        acc = tmax(acc, tmin(tabs(dd), tabs(uu - vn)));
                    } else {
                        const T g_lo = ub::fma_(uu, huge, -(m.lo[j] * huge)), g_hi = ub::fma_(-huge, uu, m.hi[j] * huge);
                        acc = tmax(acc, tabs((T)med3_neglo((float)dd, (float)g_hi, (float)g_lo)));
                    }
                    x[q] = m.template project<true>(ub::fma_(beta, vn - vi[q], vn), j);          // mpc.h:343 (difference form)
                    vo[q] = vn;
                }
            }
            const int go = group_or<G>(tmax(acc0, acc1) >= geps ? 1 : 0);
            ++wave_iters;
            ++iter;
            const bool ends = go == 0 || iter >= kn.max_iter;                                    // mpc.h:310-311, :271
            if (ballot_b(have && ends) != 0ull) {
                const bool stop = have && go == 0;
                const bool cap = have && !stop && iter >= kn.max_iter;
                if (stop) {   // what the group had BEFORE this iteration's update (dlib breaks before updating)
                    if constexpr (STATE) publish(xs, vi, iter - 1); else publish(x0_prev, vi, iter - 1);
                    have = false;
                }
                if (cap) { flags |= 0x2u; publish(x, vo, iter); have = false; }
                const unsigned long long waiting = ballot_b(!have && !exhausted);
                if (__popcll(waiting) >= GroupRefillBatch<G>::value * G || ballot_b(have) == 0ull) return true;
            }
            return false;
        };
#pragma unroll 1
        do {
            if (iteration(v, v2)) {
#pragma unroll
                for (int q = 0; q < NV; ++q) v[q] = v2[q];
                break;
            }
            if (iteration(v2, v)) break;
        } while (true);
    }
    raise_flags(g.flags, flags);
    if (stats && lane == 0) {
        atomicAdd(&stats[0], (unsigned long long)wave_iters);
        atomicAdd(&stats[1], (unsigned long long)refills);
    }
}

}  // namespace tpc
