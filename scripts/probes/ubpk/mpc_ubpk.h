// LANE_FMA, fp32, TWO instances per lane on packed math (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32): the projected-
// gradient kernel of the compact form for the batches that pass both stop-test screens (ub_pg_kernel's MODE 2).
//
// Why.  profiles/r04_fp32_issue.txt: at two wavefronts per SIMD the fp32 kernel issues one VALU instruction per 3.5-4
// cycles -- the unit is saturated, and what is left is the instruction count.  A packed instruction does the work of two
// for the multiply-add part (440 of an N = 20 iteration's 720 instructions when two instances share a lane; min / max /
// med3 have no packed fp32 form and stay one per instance), and costs 2.06 ns at ONE wavefront per SIMD against 1.4-1.7 ns
// per plain fused multiply-add at two.  Two instances per lane need x and v of both in the 256 registers a VALU instruction
// can name (160), so the stored forward pass goes to the other half of the register file (AGPRs: one move per dword and
// direction).
//
// What is computed is ub_pg_kernel<float, H, true, 2>'s arithmetic, operation for operation, on each half of every packed
// register (the packed instructions round each half like their scalar forms): the same bits as that kernel and as the CPU
// model (tests/model/), which the fp32 tests hold it to.  Records, queue, screens and the other two builds (mask-as-
// arithmetic, exact) are ub_cd_kernel's and ub_pg_kernel's own.  Reference: dlib_files/dlib/control/mpc.h:275-283,
// 289-311, 336-345 in the arithmetic of mpc_ub_model.h.
#pragma once

#include "../../../trajectory_controller_amd/csrc/mpc_ub.h"   // (probe: compiled by adding it to mpc_ub_inst.hip, see the header of profiles/r04_ab_f32_packed.txt)

namespace tpc {

typedef float f2 __attribute__((ext_vector_type(2)));

TPC_DEV f2 pk_fma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
TPC_DEV f2 pk_splat(float x) { f2 r; r.x = x; r.y = x; return r; }
TPC_DEV f2 pk_med3(f2 v, float lo, float hi) {
    f2 r;
    r.x = __builtin_amdgcn_fmed3f(v.x, lo, hi);
    r.y = __builtin_amdgcn_fmed3f(v.y, lo, hi);
    return r;
}
// max(acc, min(|d|, |m|)) per half (v_min_f32 with |.| modifiers, v_max_f32)
TPC_DEV f2 pk_acc(f2 acc, f2 d, f2 mv) {
    f2 r;
    r.x = __builtin_fmaxf(acc.x, __builtin_fminf(__builtin_fabsf(d.x), __builtin_fabsf(mv.x)));
    r.y = __builtin_fmaxf(acc.y, __builtin_fminf(__builtin_fabsf(d.y), __builtin_fabsf(mv.y)));
    return r;
}
TPC_DEV void agpr_put2(AgprWord& s, f2 x) {
    asm("v_accvgpr_write_b32 %0, %1" : "=a"(s.lo) : "v"(__float_as_int(x.x)));
    asm("v_accvgpr_write_b32 %0, %1" : "=a"(s.hi) : "v"(__float_as_int(x.y)));
}
TPC_DEV f2 agpr_get2(const AgprWord& s) {
    int lo, hi;
    asm("v_accvgpr_read_b32 %0, %1" : "=v"(lo) : "a"(s.lo));
    asm("v_accvgpr_read_b32 %0, %1" : "=v"(hi) : "a"(s.hi));
    f2 r; r.x = __int_as_float(lo); r.y = __int_as_float(hi);
    return r;
}

#ifndef TPC_UBPK_WV
#define TPC_UBPK_WV 0   // horizon steps of the stored forward pass kept in VGPRs (the rest in AGPRs)
#endif
#ifndef TPC_UBPK_RV
#define TPC_UBPK_RV 0   // A/B: 1 = the forward pass regenerated in the backward sweep (ub::Unit::rev) instead of stored
#endif
template <int H> struct UbPkPlan {
    static constexpr bool built = H == 20 || H == 10;
    static constexpr int wv = TPC_UBPK_WV;
    static constexpr int refill_slots = 2 * RefillBatch<H>::value;   // free slots a refill pass waits for
};

template <int H>
__global__ __launch_bounds__(64, 1) void ubpk_pg_kernel(CompactArgs g, Knobs kn, const float* __restrict__ recs,
                                                        const uint32_t* __restrict__ order, uint32_t* __restrict__ ticket,
                                                        unsigned long long* __restrict__ stats,
                                                        const uint32_t* __restrict__ queue_len) {
    using T = float;
    using P = UbPkPlan<H>;
    constexpr int RL = LaneRec<T, H>::kLen;
    constexpr int WV = P::wv, WA = H - WV;
    const int64_t n_queue = (int64_t)__builtin_nontemporal_load(queue_len);
    {   // this build serves the batches both screens admit (ub_pg_kernel's MODE 2); the other builds take the rest
        const unsigned long long sel = __builtin_nontemporal_load(&stats[2]);
        if ((sel & 3ull) != 0ull) return;
        if (n_queue <= 0) return;
    }
    const int lane = threadIdx.x;
    constexpr T gs = ub::GradScale<T>::g;
    const T geps = gs * (T)kn.eps;

    // the uniform part of the model (ub::Unit<float, true>, dlib's coordinates: s = 1, lo = 0)
    ub::Unit<T, true> mu;
    ub_set_uniform(mu, gs, g);
    const T bl0 = mu.bl0, bl1 = mu.bl1, bh0 = mu.bh0, bh1 = mu.bh1;
    const f2 gq0 = pk_splat(mu.gq0), gq1 = pk_splat(mu.gq1), grs0 = pk_splat(mu.grs0), grs1 = pk_splat(mu.grs1);
    const f2 grl0 = pk_splat(mu.grl0), grl1 = pk_splat(mu.grl1);
    // per slot
    f2 a = pk_splat(0.f), c = a, as1 = a, cs0 = a, cs1 = a, z0 = a, q1th = a, il0 = a, il1 = a, beta = a;
    f2 x[2 * H], v[2 * H];
    f2 w_v[WV > 0 ? 2 * WV : 1];
    AgprWord w_a[WA > 0 ? 2 * WA : 1];
    auto w_put = [&](int i, f2 Z, f2 Y) {
        if (i < WV) { w_v[2 * i] = Z; w_v[2 * i + 1] = Y; }
        else { agpr_put2(w_a[2 * (i - WV)], Z); agpr_put2(w_a[2 * (i - WV) + 1], Y); }
    };
    auto w_getz = [&](int i) -> f2 { if (i < WV) return w_v[2 * i]; else return agpr_get2(w_a[2 * (i - WV)]); };
    auto w_gety = [&](int i) -> f2 { if (i < WV) return w_v[2 * i + 1]; else return agpr_get2(w_a[2 * (i - WV) + 1]); };
    f2 x0_prev[2] = {pk_splat(0.f), pk_splat(0.f)};
    int64_t k0 = 0, k1 = 0;
    uint32_t iter0 = 0, iter1 = 0;
    bool have0 = false, have1 = false, exhausted = false;
    uint32_t flags = 0;
    uint32_t wave_iters = 0, refills = 0;
#pragma unroll
    for (int q = 0; q < 2 * H; ++q) { x[q] = pk_splat(0.f); v[q] = pk_splat(0.f); }
    if constexpr (TPC_UBPK_RV == 0) {
#pragma unroll
        for (int i = 0; i < H; ++i) w_put(i, pk_splat(0.f), pk_splat(0.f));
    }

    auto publish = [&](int64_t k, T a0, T a1, uint32_t it) {   // (dlib's coordinates: x is the control)
        ((T*)g.front)[k] = a0;
        ((T*)g.rear)[k] = a1;
        if (g.iters) g.iters[k] = (int32_t)it;
    };
    // one slot's refill from its record (left by ub_cd_kernel; see ub_pg_kernel); returns whether the slot now iterates
    auto load_slot = [&](int64_t k, auto set_x, auto set_v, auto set_model, uint32_t& iter) -> bool {
        const T* rec = recs + k * RL;
        const T* ex = rec + LaneRec<T, H>::kExtra;
        T xs[2 * H];
#pragma unroll
        for (int q = 0; q < 2 * H; ++q) xs[q] = rec[q];
        const uint64_t meta = load_meta<T>(rec + 2 * H + 1);
        iter = (uint32_t)meta;
        if (meta & kMetaNonFinite) flags |= 0x1u;
        const bool vinit = (meta & kMetaVInit) != 0;   // mpc.h:330-334, else a fresh v = 0
#pragma unroll
        for (int q = 0; q < 2 * H; ++q) { set_x(q, xs[q]); set_v(q, vinit ? xs[q] : mu.xz(q & 1)); }
        // ub::Unit::set_instance_ac with s = 1, lo = 0, in its own operations
        const T ai = ex[3], ci = ex[4], ty = ex[5], tphi = ex[6];
        set_model(ai, ci, ai * mu.s1, ci * mu.s0, ci * mu.s1, (T)0 - ty, mu.gq1 * (mu.lo1 + tphi), ex[0], ex[1], ex[2]);
        if ((meta & kMetaStopped) || iter >= kn.max_iter) {
            if (!(meta & kMetaStopped)) flags |= 0x2u;
            if (meta & kMetaNonFinite) { ((T*)g.front)[k] = (T)0; ((T*)g.rear)[k] = (T)0; if (g.iters) g.iters[k] = (int32_t)iter; }
            else publish(k, xs[0], xs[1], iter);
            return false;
        }
        return true;
    };

#pragma unroll 1
    while (true) {
        // ---- refill: free slots take the next entries of the longest-first queue (slot 0 of every lane first)
        const unsigned long long want0 = __ballot(!have0 && !exhausted), want1 = __ballot(!have1 && !exhausted);
        const int nwant = __popcll(want0) + __popcll(want1);
        if (nwant != 0 && (nwant >= P::refill_slots || (__ballot(have0) | __ballot(have1)) == 0ull)) {
            ++refills;
            const unsigned long long below = (1ull << lane) - 1ull;
            uint32_t first_ticket = 0;
            const int leader = __ffsll((long long)(want0 | want1)) - 1;
            if (lane == leader) first_ticket = atomicAdd(ticket, (uint32_t)nwant);
            first_ticket = (uint32_t)__shfl((int)first_ticket, leader);
            const uint32_t t0 = first_ticket + (uint32_t)__popcll(want0 & below);
            const uint32_t t1 = first_ticket + (uint32_t)__popcll(want0) + (uint32_t)__popcll(want1 & below);
            if (!have0 && !exhausted) {
                if ((int64_t)t0 >= n_queue) {
                    exhausted = true;
                } else {
                    k0 = (int64_t)order[t0];
                    have0 = load_slot(k0, [&](int q, T val) { x[q].x = val; }, [&](int q, T val) { v[q].x = val; },
                                      [&](T a_, T c_, T as1_, T cs0_, T cs1_, T z0_, T q1_, T i0_, T i1_, T b_) {
                                          a.x = a_; c.x = c_; as1.x = as1_; cs0.x = cs0_; cs1.x = cs1_; z0.x = z0_; q1th.x = q1_;
                                          il0.x = i0_; il1.x = i1_; beta.x = b_;
                                      }, iter0);
                }
            }
            if (!have1 && !exhausted) {
                if ((int64_t)t1 >= n_queue) {
                    exhausted = true;
                } else {
                    k1 = (int64_t)order[t1];
                    have1 = load_slot(k1, [&](int q, T val) { x[q].y = val; }, [&](int q, T val) { v[q].y = val; },
                                      [&](T a_, T c_, T as1_, T cs0_, T cs1_, T z0_, T q1_, T i0_, T i1_, T b_) {
                                          a.y = a_; c.y = c_; as1.y = as1_; cs0.y = cs0_; cs1.y = cs1_; z0.y = z0_; q1th.y = q1_;
                                          il0.y = i0_; il1.y = i1_; beta.y = b_;
                                      }, iter1);
                }
            }
        }
        if ((__ballot(have0) | __ballot(have1)) == 0ull) {
            if (__ballot(!exhausted) == 0ull) break;
            continue;
        }

#pragma unroll 1
        do {
            // ---- forward pass (ub::Unit::fwd_init / fwd)
            f2 Z = z0, Y = pk_splat(mu.lo1);
#pragma unroll
            for (int i = 0; i < H; ++i) {
                const f2 zn = pk_fma(as1, x[2 * i + 1], pk_fma(a, Y, Z));
                const f2 yn = pk_fma(cs0, x[2 * i], pk_fma(-cs1, x[2 * i + 1], Y));
                Z = zn; Y = yn;
                if constexpr (TPC_UBPK_RV == 0) w_put(i, Z, Y);
            }
            // ---- backward pass fused with the stop test and the speculative update
            x0_prev[0] = x[0]; x0_prev[1] = x[1];
            constexpr int NA = 4;
            f2 acc[NA];
#pragma unroll
            for (int z = 0; z < NA; ++z) acc[z] = pk_splat(0.f);
            f2 n0 = gq0 * Z;                                                        // ub::Unit::bwd_last
            f2 n1 = pk_fma(gq1, Y, -q1th);
            static_for<H>([&](auto ic) {
                constexpr int i = H - 1 - decltype(ic)::value;
                if constexpr (i < H - 1) {                                          // ub::Unit::bwd
                    f2 Zi, Yi;
                    if constexpr (TPC_UBPK_RV != 0) { Zi = Z; Yi = Y; } else { Zi = w_getz(i); Yi = w_gety(i); }
                    const f2 e1 = pk_fma(gq1, Yi, -q1th);
                    const f2 t1 = pk_fma(a, n0, n1) + e1;
                    n0 = pk_fma(gq0, Zi, n0);
                    n1 = t1;
                }
                if constexpr (TPC_UBPK_RV != 0 && i > 0) {                          // ub::Unit::rev: (Z, Y) of step i-1
                    const f2 yp = pk_fma(cs1, x[2 * i + 1], pk_fma(-cs0, x[2 * i], Y));
                    Z = pk_fma(-a, yp, pk_fma(-as1, x[2 * i + 1], Z));
                    Y = yp;
                    asm volatile("" : "+v"(Z), "+v"(Y));
                }
                static_for<2>([&](auto jc) {
                    constexpr int j = decltype(jc)::value;
                    constexpr int q = 2 * i + j;
                    const f2 xx = x[q];
                    f2 dd;
                    if constexpr (j == 0) dd = pk_fma(c, n1, pk_fma(grs0, xx, grl0));                      // df0
                    else dd = pk_fma(a, n0, pk_fma(-c, n1, pk_fma(grs1, xx, grl1)));                       // df1
                    const f2 vn = pk_med3(pk_fma(j == 0 ? -il0 : -il1, dd, xx), j == 0 ? bl0 : bl1, j == 0 ? bh0 : bh1);   // mpc.h:342
                    acc[q % NA] = pk_acc(acc[q % NA], dd, xx - vn);
                    x[q] = pk_med3(pk_fma(beta, vn - v[q], vn), j == 0 ? bl0 : bl1, j == 0 ? bh0 : bh1);   // mpc.h:343
                    v[q] = vn;
                });
            });
            f2 max_df = acc[0];
#pragma unroll
            for (int z = 1; z < NA; ++z) {
                max_df.x = __builtin_fmaxf(max_df.x, acc[z].x);
                max_df.y = __builtin_fmaxf(max_df.y, acc[z].y);
            }
            ++wave_iters;
            const bool stop0 = have0 && (max_df.x < geps), stop1 = have1 && (max_df.y < geps);   // mpc.h:310-311
            ++iter0; ++iter1;
            const bool cap0 = have0 && !stop0 && iter0 >= kn.max_iter, cap1 = have1 && !stop1 && iter1 >= kn.max_iter;   // mpc.h:271
            if (__ballot(stop0 || cap0 || stop1 || cap1) != 0ull) {
                if (stop0) { publish(k0, x0_prev[0].x, x0_prev[1].x, iter0 - 1); have0 = false; }
                if (cap0) { flags |= 0x2u; publish(k0, x[0].x, x[1].x, iter0); have0 = false; }
                if (stop1) { publish(k1, x0_prev[0].y, x0_prev[1].y, iter1 - 1); have1 = false; }
                if (cap1) { flags |= 0x2u; publish(k1, x[0].y, x[1].y, iter1); have1 = false; }
                const int waiting = __popcll(__ballot(!have0 && !exhausted)) + __popcll(__ballot(!have1 && !exhausted));
                if (waiting >= P::refill_slots || (__ballot(have0) | __ballot(have1)) == 0ull) break;
            }
        } while (true);
    }
    raise_flags(g.flags, flags);
    if (stats && lane == 0) {
        atomicAdd(&stats[0], (unsigned long long)wave_iters * 2ull);   // (in units of 64 instance-iterations, like the other builds)
        atomicAdd(&stats[1], (unsigned long long)refills);
    }
}

}  // namespace tpc
