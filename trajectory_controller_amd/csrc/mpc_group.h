// GROUP kernels: G lanes per MPC instance -- the family between "one wavefront per instance" (WAVE, mpc_wave.h)
// and "one lane per instance" (LANE_FMA, mpc_ub.h), for batches too small to give every lane of the chip an
// instance of its own and too large for a wavefront each.
//
// What is computed is the LANE_FMA family's projected-gradient phase (reference: accelerated projected gradient,
// dlib_files/dlib/control/mpc.h:336-345, on the gradient of mpc.h:275-283, stop test mpc.h:289-311), in the
// arithmetic of mpc_ub_model.h (unit-box coordinates, fused multiply-adds, the linear term folded into the backward
// recurrence).  The coordinate-descent phase (mpc.h:319-335), the records it leaves, the longest-first queue and the
// screen of the select-free stop test are LANE_FMA's own (ub_cd_kernel, mpc_sort.hip): this file adds one kernel,
// group_pg_kernel, that consumes the same records.
//
// Layout.  The H horizon steps of an instance are cut into G chunks of L = H / G consecutive steps; lane p of a
// group owns chunk p: its 2L controls x, dlib's momentum v, and the chunk's part of both recurrences, all in
// registers.  A wavefront carries 64 / G instances.  Both recurrences of mpc.h:275-281 are affine in the state they
// carry with a CONSTANT matrix (A = [1 a; 0 1], so A^k = [1 ka; 0 1]), hence each is computed as
//     1. the chunk's recurrence from a zero start (L steps in registers, every lane at once),
//     2. an exclusive scan over the G lanes of the chunk summaries -- a summary is the chunk's offset pair, and
//        combining with the summary d lanes away is `z += z' + (d L a) y';  y += y'`: log2 G steps of two
//        DPP-moved values and three fused multiply-adds, out-of-group lanes cancelled by 0 / 1 weights --
//     3. the correction of the L local values by the incoming state (an arithmetic progression: one addition each).
// Where G does not divide H the chunks are L = ceil(H / G) steps and the last ones are padded with dummy steps that
// cost nothing in the objective and never move (GroupPlan).
// Forward: (Z, Y) = predicted state error;  backward: (N0, N1) = the costate.  The stop test (the largest free
// gradient component against eps) is a per-lane maximum and one OR over the group, the arg-max is not needed in this
// phase.  30 VALU instructions per horizon step against LANE_FMA's 25, plus ~36 (G = 4) for the two scans: at N = 20
// an iteration is ~200 instructions for 16 instances where WAVE spends ~90 on one or two and LANE_FMA 660 on 64 that
// must all be there.
//
// Lanes of a group finish together (the verdict is the group's), and a group whose instance has finished takes the
// next one from the longest-first queue exactly as a LANE_FMA lane does (one ticket atomic per refill pass).
//
// Results: the same decisions as dlib on quantities that differ from dlib's by rounding (a third association of the
// same sums, beside WAVE's and LANE_FMA's): held to <= 1e-9 and to identical iteration counts against the oracle
// and the real-dlib fixtures (tests/test_group_gpu.py).  Only the screened ("fast") stop test is built; a batch the
// screen refuses runs LANE_FMA's exact build on the same records.
#pragma once

#include "mpc_ub.h"

namespace tpc {

// DPP moves inside a group of G lanes: the value of the lane `D` places below (Shr) or above (Shl).  Where that lane
// lies outside the group the result is another lane's value or zero -- finite either way (only screened, finite
// instances are ever loaded), and multiplied by a zero weight where it is used.
template <int G, int D, bool UP> struct GroupDpp {
    // quad_perm codes keep every read inside the quad; wider groups use row shifts with zero fill
    static constexpr int ctrl =
        G <= 4 ? (UP ? (D == 1 ? (G == 2 ? 0xF5 : 0xF9) : 0xEE)     // [1,1,3,3] / [1,2,3,3] / [2,3,2,3]
                     : (D == 1 ? (G == 2 ? 0xA0 : 0x90) : 0x44))    // [0,0,2,2] / [0,0,1,2] / [0,1,0,1]
               : (UP ? 0x100 + D : 0x110 + D);                      // row_shl:D / row_shr:D
};
template <int CTRL> TPC_DEV double group_mov(double x) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
template <int CTRL> TPC_DEV float group_mov(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, true));
}
// OR over the lanes of a group, left in every lane (butterfly: quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror)
template <int G> TPC_DEV int group_or(int x) {
    if constexpr (G > 1) x |= __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xf, 0xf, true);
    if constexpr (G > 2) x |= __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xf, 0xf, true);
    if constexpr (G > 4) x |= __builtin_amdgcn_update_dpp(0, x, 0x141, 0xf, 0xf, true);
    if constexpr (G > 8) x |= __builtin_amdgcn_update_dpp(0, x, 0x140, 0xf, 0xf, true);
    return x;
}

// ballot of a predicate as the compare that produced it leaves it (the int-typed __ballot re-materialises it)
TPC_DEV unsigned long long ballot_b(bool pred) { return __builtin_amdgcn_ballot_w64(pred); }

#ifndef TPC_GROUP_OCC
#define TPC_GROUP_OCC 0
#endif
#ifndef TPC_GROUP_OCC_F32
#define TPC_GROUP_OCC_F32 2
#endif
template <typename T, int H, int G> struct GroupPlan {
    static_assert(G == 2 || G == 4 || G == 8 || G == 16, "a group is 2, 4, 8 or 16 lanes of one DPP row");
    static constexpr int L = (H + G - 1) / G;  // horizon steps per lane
    // Where G does not divide H the last chunks are padded with DUMMY steps (global index >= H): their controls rest at
    // u = 0, their stage cost is zero (nothing enters the backward recurrence) and their gradient is multiplied by zero
    // (no step, no say in the stop test).  What a dummy step does to the forward state is seen by no real step.
    static constexpr int pad = G * L - H;
    static constexpr int dl0 = pad >= L ? 0 : L - pad;   // local steps dl0 .. L-1 are dummy in some lane
    static constexpr int NG = kWave / G;       // instances per wavefront
    // scan steps after the initial shift by one lane: the exclusive prefix of lane p spans up to G - 1 lanes, and each
    // step doubles what a lane's partial result spans (1 after the shift)
    static constexpr int steps = G == 2 ? 0 : (G == 4 ? 2 : (G == 8 ? 3 : 4));
    // fp64: ONE wavefront per SIMD, like LANE_FMA: the family's place is the batch that cannot fill the chip's lanes, where
    // what counts is how fast a lone wavefront iterates -- measured (N = 20, G = 4, 16 384 / 32 768 / 65 536 instances,
    // persistent grids of one / two / three wavefronts per SIMD): 1.65 / 1.73 / 2.42 ms against 1.64 / 1.82 / 2.50 and
    // 1.63 / 1.87 / 2.87 -- so the kernel is built with the whole register file and no scratch.
    // fp32: built for TWO (256 registers each; a second wavefront per SIMD brings an fp32 instruction from ~2.0 to ~1.2-1.5 ns
    // where it brings an fp64 one from 2.36 to 2.1), and the launcher pairs them only from the batch size at which the
    // grid's throughput, not the longest instance, sets the time (auto_table.h, pair_from): profiles/r04_group_f32_occ.txt --
    // 262 144 x N = 20: G = 2 6.24 -> 4.56 ms, N = 40: G = 4 53.0 -> 34.3; three and four per SIMD spill at N >= 30.
    static constexpr int occ = TPC_GROUP_OCC > 0 ? TPC_GROUP_OCC : (sizeof(T) == 4 ? TPC_GROUP_OCC_F32 : 1);
};
// groups that wait for an instance before a refill pass is worth leaving the loop for
template <int G> struct GroupRefillBatch { static constexpr int value = G >= 8 ? 1 : 2; };

// MOVED (fp32 only; fp64 always reads its stop test off the projected step): true = min(|g df|, |x - x_new|) where the
// coordinate-descent kernel's second screen (ub::moved_stop_ok, stats[2] bit 1) allows it, false = dlib's mask as
// arithmetic for the batches it refuses -- the two builds of ub_pg_kernel's MODE 2 / MODE 1, launched back to back.
template <typename T, int H, int G, bool EQB, bool MOVED = true>
__global__ __launch_bounds__(64, (GroupPlan<T, H, G>::occ)) void group_pg_kernel(
    CompactArgs g, Knobs kn, const T* __restrict__ recs, const uint32_t* __restrict__ order,
    uint32_t* __restrict__ ticket, unsigned long long* __restrict__ stats, const uint32_t* __restrict__ queue_len) {
    using P = GroupPlan<T, H, G>;
    constexpr int L = P::L, RL = LaneRec<T, H>::kLen;
    constexpr bool D64 = sizeof(T) == 8;
    const int64_t n_queue = (int64_t)__builtin_nontemporal_load(queue_len);
    // the screened stop test only: a batch the coordinate-descent kernel's screen refused is LANE_FMA's exact build's
    static_assert(MOVED || sizeof(T) == 4, "the mask-as-arithmetic build is fp32's");
    {
        const unsigned long long sel = __builtin_nontemporal_load(&stats[2]);
        if ((sel & 1ull) != 0ull) return;
        if constexpr (!D64) { if (((sel & 2ull) != 0ull) == MOVED) return; }   // (bit 1: a step may vanish in fp32 rounding -- the mask build's batch)
    }
    if (n_queue <= 0) return;

    const int lane = threadIdx.x;
    const int p = lane & (G - 1);                 // chunk of this lane
    const int gbase = lane & ~(G - 1);            // first lane of its group
    constexpr T gs = ub::GradScale<T>::g;
    const T geps = gs * (T)kn.eps;
    T huge = (T)0x1p100;                          // fp32: dlib's mask as arithmetic (ub_pg_kernel)
    asm volatile("" : "+v"(huge));

    ub::Unit<T, EQB> m;
    ub_set_uniform(m, gs, g);
    m.a = m.c = m.as1 = m.cs0 = m.cs1 = m.dlt = m.z0 = m.q1th = (T)0;
    // 0 / 1 weights of the scan steps: lane p combines with the lane d below (forward) / above (backward) if that
    // lane belongs to the group
    constexpr int KS = P::steps, KA = KS > 0 ? KS : 1;
    T wf[KA], wb[KA];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        wf[s] = p >= (1 << s) ? (T)1 : (T)0;
        wb[s] = p + (1 << s) < G ? (T)1 : (T)0;
    }
    const T wf0 = p >= 1 ? (T)1 : (T)0, wb0 = p + 1 < G ? (T)1 : (T)0;   // the initial shift by one lane
    // 1 for a real step, 0 for a dummy one (local steps DL0 .. L-1 only; the others are real in every lane), and the
    // stage weights of those steps
    constexpr int DL0 = P::dl0, ND = L - DL0 > 0 ? L - DL0 : 1;
    T live[ND], gq0d[ND], gq1d[ND];
#pragma unroll
    for (int l = DL0; l < L; ++l) {
        live[l - DL0] = p * L + l < H ? (T)1 : (T)0;
        gq0d[l - DL0] = m.gq0 * live[l - DL0];
        gq1d[l - DL0] = m.gq1 * live[l - DL0];
    }
    // per instance: d L a times the weight, and the start of the local forward pass (the true start in chunk 0)
    T cf[KA], cb[KA];
#pragma unroll
    for (int s = 0; s < KS; ++s) cf[s] = cb[s] = (T)0;
    T zst = (T)0, yst = (T)0, la = (T)0;

    T x[2 * L], v[2 * L], v2[2 * L];
    T x0_prev[2] = {(T)0, (T)0};
    T il[2] = {(T)0, (T)0}, beta = (T)0;
    int64_t k = 0;
    uint32_t iter = 0;
    bool have = false, exhausted = false;        // (the same in every lane of a group)
    uint32_t flags = 0;
    uint32_t wave_iters = 0, refills = 0;
#pragma unroll
    for (int q = 0; q < 2 * L; ++q) { x[q] = (T)0; v[q] = (T)0; v2[q] = (T)0; }

    auto publish = [&](T a0, T a1, uint32_t it) {   // chunk 0 holds step 0
        if (p == 0) {
            ((T*)g.front)[k] = m.control(0, a0);
            ((T*)g.rear)[k] = m.control(1, a1);
            if (g.iters) g.iters[k] = (int32_t)it;
        }
    };

#pragma unroll 1
    while (true) {
        // ---- refill: groups without an instance take the next entries of the longest-first queue
        const unsigned long long want = ballot_b(!have && !exhausted);
        if (want != 0ull && (__popcll(want) >= GroupRefillBatch<G>::value * G || ballot_b(have) == 0ull)) {
            ++refills;
            const uint32_t cnt = (uint32_t)__popcll(want) / G;
            const uint32_t rank = (uint32_t)__popcll(want & ((1ull << gbase) - 1ull)) / G;
            const int leader = __ffsll((long long)want) - 1;
            uint32_t first_ticket = 0;
            if (lane == leader) first_ticket = atomicAdd(ticket, cnt);
            first_ticket = (uint32_t)__shfl((int)first_ticket, leader);
            if (!have && !exhausted) {
                const uint32_t t = first_ticket + rank;
                if ((int64_t)t >= n_queue) {
                    exhausted = true;
                } else {
                    k = (int64_t)order[t];
                    const T* rec = recs + k * RL;
                    const T* ex = rec + LaneRec<T, H>::kExtra;   // (left by ub_cd_kernel: step constants, a, c, target)
#pragma unroll
                    for (int q = 0; q < 2 * L; ++q) {
                        if (q / 2 < DL0) x[q] = rec[2 * L * p + q];
                        else x[q] = p * L + q / 2 < H ? rec[2 * L * p + q] : m.xz(q & 1);
                    }
                    const uint64_t meta = load_meta<T>(rec + 2 * H + 1);
                    iter = (uint32_t)meta;
                    const bool vinit = (meta & kMetaVInit) != 0;   // mpc.h:330-334, else a fresh v = 0
#pragma unroll
                    for (int q = 0; q < 2 * L; ++q) v[q] = vinit ? x[q] : m.xz(q & 1);
                    m.set_instance_ac(ex[3], ex[4], ex[5], ex[6]);
                    T dummy_z, dummy_y;
                    m.fwd_init(dummy_z, dummy_y);
                    zst = p == 0 ? dummy_z : (T)0;
                    yst = p == 0 ? dummy_y : (T)0;
                    la = (T)L * m.a;
                    if constexpr (G == 4) {   // (the shortcut scan: one chunk of L steps sits behind the moved pair)
                        cf[0] = wf[1] * la;
                        cb[0] = wb[1] * la;
                    } else {
#pragma unroll
                        for (int s = 0; s < KS; ++s) {
                            const T dla = (T)((1 << s) * L) * m.a;
                            cf[s] = wf[s] * dla;
                            cb[s] = wb[s] * dla;
                        }
                    }
                    if ((meta & kMetaStopped) || iter >= kn.max_iter) {
                        // (ub_cd_kernel publishes these itself and keeps them out of the queue; kept for a queue that holds one)
                        if (!(meta & kMetaStopped)) flags |= 0x2u;
                        if (meta & kMetaNonFinite) { flags |= 0x1u; if (p == 0) { ((T*)g.front)[k] = (T)0; ((T*)g.rear)[k] = (T)0; if (g.iters) g.iters[k] = (int32_t)iter; } }
                        else publish(x[0], x[1], iter);
                    } else {
                        il[0] = ex[0]; il[1] = ex[1]; beta = ex[2];                    // mpc.h:342-343 (ub::pg_constants)
                        have = true;
                    }
                }
            }
        }
        if (ballot_b(have) == 0ull) {
            if (ballot_b(!exhausted) == 0ull) break;
            continue;
        }

        // One iteration.  dlib's momentum vector is read from `vi` and written to `vo`: the loop below alternates two
        // arrays, because an array updated in place costs a register copy per element at the loop's back edge (the new v
        // is defined while the old one is still needed by the momentum step, so the two cannot share a register).
        // Returns true when the loop must be left (a refill pass is due, or no group has an instance any more).
        auto iteration = [&](T (&vi)[2 * L], T (&vo)[2 * L]) -> bool {
            bool stop = false, cap = false;
            // ---- forward recurrence of the chunk from its local start (mpc.h:275-277)
            T wz[L], wy[L];
            T Z = zst, Y = yst;
#pragma unroll
            for (int l = 0; l < L; ++l) {
                m.fwd(Z, Y, x[2 * l], x[2 * l + 1]);
                wz[l] = Z; wy[l] = Y;
            }
            // ---- exclusive scan of the chunk summaries over the group: (ez, ey) = state entering this chunk
            //      minus what the local start already carried
            T ez = wf0 * group_mov<GroupDpp<G, 1, false>::ctrl>(Z), ey = wf0 * group_mov<GroupDpp<G, 1, false>::ctrl>(Y);
            if constexpr (G == 4) {
                // four lanes: the pair (lane below, this lane) combined once, moved two lanes up and put in front of the
                // single lane below -- one move fewer than shifting first and doubling twice
                const T iz = ub::fma_(la, ey, ez + Z), iy = ey + Y;
                const T oz = group_mov<GroupDpp<G, 2, false>::ctrl>(iz), oy = group_mov<GroupDpp<G, 2, false>::ctrl>(iy);
                ez = ub::fma_(cf[0], oy, ub::fma_(wf[1], oz, ez));
                ey = ub::fma_(wf[1], oy, ey);
            } else {
                static_for<KS>([&](auto sc) {
                    constexpr int s = decltype(sc)::value;
                    constexpr int ctrl = GroupDpp<G, (1 << s), false>::ctrl;
                    const T oz = group_mov<ctrl>(ez), oy = group_mov<ctrl>(ey);
                    ez = ub::fma_(cf[s], oy, ub::fma_(wf[s], oz, ez));
                    ey = ub::fma_(wf[s], oy, ey);
                });
            }
            // ---- backward recurrence of the chunk on the corrected states, from a zero costate (mpc.h:278-281)
            x0_prev[0] = x[0]; x0_prev[1] = x[1];
            const T aey = m.a * ey;
            const T e1c = ub::fma_(m.gq1, ey, -m.q1th);
            T zc = ub::fma_((T)L, aey, ez);
            T nl0[L], nl1[L];
            T n0 = (T)0, n1 = (T)0;
#pragma unroll
            for (int l = L - 1; l >= 0; --l) {
                const T zt = wz[l] + zc;
                if (l > 0) zc = zc - aey;
                T e1;
                if (l >= DL0) e1 = ub::fma_(gq1d[l - DL0], wy[l], e1c * live[l - DL0]); else e1 = ub::fma_(m.gq1, wy[l], e1c);
                const T t1 = ub::fma_(m.a, n0, n1) + e1;
                if (l >= DL0) n0 = ub::fma_(gq0d[l - DL0], zt, n0); else n0 = ub::fma_(m.gq0, zt, n0);
                n1 = t1;
                nl0[l] = n0; nl1[l] = n1;
            }
            // ---- exclusive suffix scan: (f0, f1) = costate at the first step of the next chunk
            T f0 = wb0 * group_mov<GroupDpp<G, 1, true>::ctrl>(n0), f1 = wb0 * group_mov<GroupDpp<G, 1, true>::ctrl>(n1);
            if constexpr (G == 4) {
                const T i1 = ub::fma_(la, f0, f1 + n1), i0 = f0 + n0;
                const T o0 = group_mov<GroupDpp<G, 2, true>::ctrl>(i0), o1 = group_mov<GroupDpp<G, 2, true>::ctrl>(i1);
                f1 = ub::fma_(cb[0], o0, ub::fma_(wb[1], o1, f1));
                f0 = ub::fma_(wb[1], o0, f0);
            } else {
                static_for<KS>([&](auto sc) {
                    constexpr int s = decltype(sc)::value;
                    constexpr int ctrl = GroupDpp<G, (1 << s), true>::ctrl;
                    const T o0 = group_mov<ctrl>(f0), o1 = group_mov<ctrl>(f1);
                    f1 = ub::fma_(cb[s], o0, ub::fma_(wb[s], o1, f1));
                    f0 = ub::fma_(wb[s], o0, f0);
                });
            }
            // ---- gradient (mpc.h:283), stop test (mpc.h:289-311) and the speculative update (mpc.h:342-343)
            //      (the costate's first component enters df through a N0 only: a f0 joins the constant term)
            const T af0 = m.a * f0;
            const T grl1f = af0 + m.grl1;
            T n1c = ub::fma_((T)L, af0, f1);
            T acc0 = (T)0, acc1 = (T)0;
#pragma unroll
            for (int l = 0; l < L; ++l) {
                const T N1 = nl1[l] + n1c;
                if (l + 1 < L) n1c = n1c - af0;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int q = 2 * l + j;
                    const T xx = x[q];
                    T dd = j == 0 ? m.df0(N1, xx) : ub::fma_(m.a, nl0[l], ub::fma_(-m.c, N1, ub::fma_(m.grs1, xx, grl1f)));
                    if (l >= DL0) dd = dd * live[l - DL0];
                    const T vn = m.template project<true>(ub::fma_(-il[j], dd, xx), j);          // mpc.h:342
                    T& acc = j == 0 ? acc0 : acc1;
                    if constexpr (D64 || MOVED) {
                        acc = tmax(acc, tmin(tabs(dd), tabs(xx - vn)));
                    } else {
                        const T g_lo = m.gap_lo(j, xx, huge), g_hi = m.gap_hi(j, xx, huge);
                        acc = tmax(acc, tabs((T)med3_neglo((float)dd, (float)g_hi, (float)g_lo)));
                    }
                    x[q] = m.template project<true>(ub::fma_(beta, vn - vi[q], vn), j);          // mpc.h:343 (difference form)
                    vo[q] = vn;
                }
            }
            const int go = group_or<G>(tmax(acc0, acc1) >= geps ? 1 : 0);
            ++wave_iters;
            ++iter;
            const bool ends = go == 0 || iter >= kn.max_iter;                       // mpc.h:310-311, :271
            if (ballot_b(have && ends) != 0ull) {
                stop = have && go == 0;
                cap = have && !stop && iter >= kn.max_iter;
                if (stop) { publish(x0_prev[0], x0_prev[1], iter - 1); have = false; }
                if (cap) { flags |= 0x2u; publish(x[0], x[1], iter); have = false; }
                const unsigned long long waiting = ballot_b(!have && !exhausted);
                if (__popcll(waiting) >= GroupRefillBatch<G>::value * G || ballot_b(have) == 0ull) return true;
            }
            return false;
        };
#pragma unroll 1
        do {
            if (iteration(v, v2)) {
#pragma unroll
                for (int q = 0; q < 2 * L; ++q) v[q] = v2[q];   // (outside this loop `v` is the current array)
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
