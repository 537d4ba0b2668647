// LANEX: the BIT-EXACT projected-gradient phase with G lanes per instance -- the kernel that solves, once more and in
// dlib's own arithmetic, the instances a tolerance family left on the iteration cap (AUTO's guarantee, tpc_mpc_api.cpp:
// wants_cap_resolve; lane_cd_kernel<RESOLVE> queues them).  Those instances run max_iter iterations each, there are few of
// them (a tenth of an N = 40 batch), and with one lane per instance (lane_pg_fused_kernel) the pass lasts max_iter times one
// wavefront's ~2 200-instruction iteration: 55 ms at N = 40 with dlib's cap of 10 000.  This kernel shortens the iteration,
// not the count.
//
// What may be shared out without touching a bit.  dlib's iteration (dlib_files/dlib/control/mpc.h:275-283, 289-311, 342-343)
// in the compact model's operation order (mpc_model.h, CompactModel::first / fwd / bwd / btm) is, per horizon step,
//     8  operations ON the two recurrences' dependent chains   (forward  m0' = (m0 + a m1) + e,  m1' = m1 + d;
//                                                                backward n1' = p1 + (a n0 + n1),  n0' = p0 + n0)
//    ~41 operations OFF them: the increments e = a u1, d = c u0 - c u1 and p = w .* Q, the gradient (mpc.h:283), dlib's mask
//        and maximum (mpc.h:298-309), the projected step and the momentum step (mpc.h:342-343).
// A scan would reassociate the chains, so they stay sequential -- but lane p of a group owns the L = 5 consecutive steps of
// chunk p and does the off-chain work for those only.  The chains are HANDED from chunk to chunk: in round r every lane
// runs its chunk's 4 L chain operations from the state its lower (forward) / upper (backward) neighbour held after round
// r - 1.  Lane 0's input is fixed, so its chunk is right from round 0 on, lane 1's from round 1 on, ... after G rounds every
// lane holds dlib's values: no masks, no selects on the results, the same IEEE operations on the same operands as one lane
// doing all H steps.  The chain's start is written as an ordinary step too: from the state (-0, -0) and with |a| in the
// first product, (-0 + |a| (-0)) + e = e and -0 + d = d bit for bit (signed zeros included), which is CompactModel::first,
// and likewise p0 + -0, p1 + (|a| (-0) + -0) at the last step of the backward pass.
// Per iteration at N = 40 (G = 8): 2 x 8 rounds x (20 chain operations + 8 moves and selects for the hand-off) + 5 steps x
// ~47 + the group's verdict = ~720 instructions against ~2 200: 1.9 us instead of 5.5.  N = 20 (G = 4): ~500 against 1 096.
//
// Records, queue and refill are lane_pg_fused_kernel's (a group refills like a lane there); persistent, one wavefront per SIMD.
// fp64 only (the re-solve exists to deliver dlib's bits), N = 10, 20, 30, 40 (chunks of five steps; at N = 30 six lanes of a
// group of eight hold a chunk and six rounds hand the chains through).
// Second user: the LANE family itself on batches that cannot fill the chip one lane per instance (mpc_lane_inst.hip, run):
// there the same trade as GROUP's against LANE_FMA -- a third of the latency for a third of the full-chip throughput.
#pragma once

#include <type_traits>

#include "mpc_group.h"
#include "mpc_lane.h"

namespace tpc {

template <int H> struct LanexPlan {
    static constexpr int L = 5;                       // horizon steps per lane
    static constexpr int GA = H / L;                  // lanes of a group that hold a chunk
    static constexpr int G = GA <= 2 ? 2 : (GA <= 4 ? 4 : 8);   // lanes per instance (a power of two: the DPP moves); N = 30: 6 of 8 work
    static constexpr bool built = H == 10 || H == 20 || H == 30 || H == 40;
    static constexpr int NG = built ? kWave / G : 1;  // instances per wavefront
};

// SOLO (AUTO's presolve, tpc_mpc_api.cpp): the kernel runs BESIDE a tolerance family's persistent grid.  Sharing a SIMD with
// that grid's wavefronts more than doubles this kernel's 10 000-iteration chain (measured: 34 ms against 14), so the SOLO
// build claims enough of the register file (a clobbered AGPR: 352 registers) that no other fp64 kernel of this library fits
// beside it, and it only runs when its whole queue fits `solo_limit` instances at once (one round of the chain); a longer
// queue makes it -- and the merge and the second pass's exclusion, which read the same word -- leave the job to the ordinary
// second pass.
template <typename T, int H, bool SOLO = false>
__global__ __launch_bounds__(64, 1) void lanex_pg_kernel(CompactArgs g, Knobs kn, const T* __restrict__ recs,
                                                         const uint32_t* __restrict__ order, uint32_t* __restrict__ ticket,
                                                         unsigned long long* __restrict__ stats,
                                                         const uint32_t* __restrict__ queue_len, int refill_groups,
                                                         uint32_t solo_limit = 0u) {
    using P = LanexPlan<H>;
    if constexpr (SOLO) {
        asm volatile("" ::: "a159");
        if (__builtin_nontemporal_load(queue_len) > solo_limit) return;
    }
    static_assert(P::built && P::GA * P::L == H, "chunks of five steps: N = 10, 20, 30, 40");
    constexpr int L = P::L, G = P::G, GA = P::GA, NG = P::NG, RL = LaneRec<T, H>::kLen;
    const int64_t n_queue = (int64_t)__builtin_nontemporal_load(queue_len);
    if (n_queue <= 0) return;
    // every queued instance fits a group of the first ceil(n_queue / NG) wavefronts at once: the others are not needed
    if ((int64_t)blockIdx.x * NG >= n_queue) return;

    const int lane = threadIdx.x;
    const int p = lane & (G - 1);                 // chunk of this lane
    const int gbase = lane & ~(G - 1);            // first lane of its group
    const bool first_chunk = p == 0, last_chunk = p == GA - 1;
    const bool active = p < GA;                   // (N = 30: lanes 6 and 7 of a group hold no chunk; they run along on zeros)
    const T eps = (T)kn.eps;
    T nz = -(T)0;                                 // the chains' start state (see above)
    asm volatile("" : "+v"(nz));

    CompactModel<T> m;
    m.a = m.c = m.ty = m.tphi = (T)0;
    m.q0 = (T)g.q[0]; m.q1 = (T)g.q[1]; m.r0 = (T)g.r[0]; m.r1 = (T)g.r[1];
    m.l0 = (T)g.lo[0]; m.l1 = (T)g.lo[1]; m.h0 = (T)g.hi[0]; m.h1 = (T)g.hi[1];
    T af = (T)0, ab = (T)0;                       // a in the first forward / backward product of the chunk (|a| at a chain's start)
    T u[2 * L], v[2 * L], v2[2 * L], mm[2 * L];
    T u0_prev[2] = {(T)0, (T)0};
    T inv_lambda = (T)0, beta = (T)0;
    int64_t k = 0;
    uint32_t iter = 0;
    bool have = false, exhausted = false;         // (the same in every lane of a group)
    uint32_t flags = 0;
    uint32_t wave_iters = 0, refills = 0;
#pragma unroll
    for (int q = 0; q < 2 * L; ++q) { u[q] = (T)0; v[q] = (T)0; v2[q] = (T)0; mm[q] = (T)0; }

    auto publish = [&](T a0, T a1, uint32_t it) {   // chunk 0 holds step 0
        if (p == 0) {
            ((T*)g.front)[k] = a0;
            ((T*)g.rear)[k] = a1;
            if (g.iters) g.iters[k] = (int32_t)it;
        }
    };

#pragma unroll 1
    while (true) {
        // ---- refill: a group without an instance takes the next entry of the queue (see lane_pg_fused_kernel)
        // (`refill_groups` groups wait before a pass is worth leaving the loop for: 1 behind the re-solve's queue of
        // max_iter-long instances, GroupRefillBatch's value when the kernel serves a whole batch)
        const unsigned long long want = ballot_b(!have && !exhausted);
        if (want != 0ull && (__popcll(want) >= refill_groups * G || ballot_b(have) == 0ull)) {
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
                    m.load(g, k);
#pragma unroll
                    for (int q = 0; q < 2 * L; ++q) u[q] = active ? rec[2 * L * (active ? p : 0) + q] : (T)0;
                    const T lambda = rec[2 * H];
                    const uint64_t meta = load_meta<T>(rec + 2 * H + 1);
                    iter = (uint32_t)meta;
                    if (meta & kMetaNonFinite) flags |= 0x1u;
                    const bool vinit = (meta & kMetaVInit) != 0;   // mpc.h:330-334, else a fresh v = 0
#pragma unroll
                    for (int q = 0; q < 2 * L; ++q) v[q] = vinit ? u[q] : (T)0;
                    if ((meta & kMetaStopped) || iter >= kn.max_iter) {
                        if (!(meta & kMetaStopped)) flags |= 0x2u;
                        publish(u[0], u[1], iter);
                    } else {
                        // the linear term (mpc.h:258-266; linear_term(CompactModel), mpc_model.h): one chain over the whole
                        // horizon, run by every lane, each keeping its chunk
                        const T w0 = ((T)0 - m.ty) * m.q0, w1 = ((T)0 - m.tphi) * m.q1;   // mpc.h:261-262
                        T n0 = w0, n1 = w1;
                        static_for<H>([&](auto ic) {
                            constexpr int i = H - 1 - decltype(ic)::value;
                            if constexpr (i < H - 1) {                                    // mpc.h:263-264
                                const T t0 = w0 + n0;
                                const T t1 = w1 + (m.a * n0 + n1);
                                n0 = t0; n1 = t1;
                            }
                            const T e0 = m.c * n1, e1 = m.a * n0 - m.c * n1;              // mpc.h:265-266
                            if (p == i / L) { mm[2 * (i % L)] = e0; mm[2 * (i % L) + 1] = e1; }
                        });
                        af = first_chunk ? tabs(m.a) : m.a;
                        ab = last_chunk ? tabs(m.a) : m.a;
                        inv_lambda = (T)1.0 / lambda;                                     // mpc.h:342
                        const T sq = tsqrt(lambda);
                        beta = (sq - (T)1) / (sq + (T)1);                                 // mpc.h:343
                        have = true;
                    }
                }
            }
        }
        if (ballot_b(have) == 0ull) {
            if (ballot_b(!exhausted) == 0ull) break;
            continue;
        }

        // One iteration; dlib's momentum vector is read from `vi` and written to `vo` (two alternating arrays: an array
        // updated in place costs a register copy per element at the loop's back edge, see group_pg_kernel).
        // Returns true when the loop must be left (a group wants an instance, or none has one any more).
        auto iteration = [&](T (&vi)[2 * L], T (&vo)[2 * L]) -> bool {
            // ---- forward pass: M[i] = A M[i-1] + B u[i]  (mpc.h:275-277), the chain handed from chunk to chunk
            T e[L], d[L], w0[L], w1[L];
#pragma unroll
            for (int l = 0; l < L; ++l) {
                e[l] = m.a * u[2 * l + 1];
                d[l] = m.c * u[2 * l] - m.c * u[2 * l + 1];
            }
            T o0 = nz, o1 = nz;   // this lane's state after its last step, as of the previous round
#pragma unroll
            for (int r = 0; r < GA; ++r) {
                T i0 = group_mov<GroupDpp<G, 1, false>::ctrl>(o0), i1 = group_mov<GroupDpp<G, 1, false>::ctrl>(o1);
                i0 = first_chunk ? nz : i0;
                i1 = first_chunk ? nz : i1;
                T m0 = (i0 + af * i1) + e[0];
                T m1 = i1 + d[0];
                w0[0] = m0; w1[0] = m1;
#pragma unroll
                for (int l = 1; l < L; ++l) {
                    const T t0 = (m0 + m.a * m1) + e[l];
                    const T t1 = m1 + d[l];
                    m0 = t0; m1 = t1;
                    w0[l] = m0; w1[l] = m1;
                }
                o0 = m0; o1 = m1;
            }
            // ---- backward pass: N = trans(A) N + Q .* M[i]  (mpc.h:278-281), handed down from the last chunk
            T p0[L], p1[L], nl0[L], nl1[L];
#pragma unroll
            for (int l = 0; l < L; ++l) { p0[l] = w0[l] * m.q0; p1[l] = w1[l] * m.q1; }
            T b0 = nz, b1 = nz;
#pragma unroll
            for (int r = 0; r < GA; ++r) {
                T i0 = group_mov<GroupDpp<G, 1, true>::ctrl>(b0), i1 = group_mov<GroupDpp<G, 1, true>::ctrl>(b1);
                i0 = last_chunk ? nz : i0;
                i1 = last_chunk ? nz : i1;
                T n0 = p0[L - 1] + i0;
                T n1 = p1[L - 1] + (ab * i0 + i1);
                nl0[L - 1] = n0; nl1[L - 1] = n1;
#pragma unroll
                for (int l = L - 2; l >= 0; --l) {
                    const T t0 = p0[l] + n0;
                    const T t1 = p1[l] + (m.a * n0 + n1);
                    n0 = t0; n1 = t1;
                    nl0[l] = n0; nl1[l] = n1;
                }
                b0 = n0; b1 = n1;
            }
            // ---- gradient (mpc.h:283), dlib's mask and maximum (mpc.h:298-309), the speculative update (mpc.h:342-343)
            u0_prev[0] = u[0]; u0_prev[1] = u[1];
            T acc = (T)0;
            auto step = [&](int l, auto test_c) {
                constexpr bool TEST = decltype(test_c)::value;
                const T cn1 = m.c * nl1[l];
                const T bt[2] = {cn1, m.a * nl0[l] - cn1};                               // CompactModel::btm
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int q = 2 * l + j;
                    const T uu = u[q];
                    const T dd = (mm[q] + bt[j]) + uu * m.R(j);                           // mpc.h:283
                    if constexpr (TEST) {
                        const T up = (uu <= m.lo(j)) ? (T)0 : dd;                         // mpc.h:298-299
                        const T dn = (uu >= m.hi(j)) ? (T)0 : -dd;
                        acc = tmax(acc, tmax(up, dn));
                    }
                    const T vn = clamp3(uu - inv_lambda * dd, m.lo(j), m.hi(j));          // mpc.h:342
                    u[q] = clamp3(vn + beta * (vn - vi[q]), m.lo(j), m.hi(j));            // mpc.h:343
                    vo[q] = vn;
                }
            };
            // The stop test decided early (as in the hand-written kernel, scripts/ubasm.py): mpc.h:310 stops only if EVERY
            // term is below eps, so a group one of whose lanes has a term >= eps after its chunk's first step goes on whatever
            // the other terms are; when that holds for every group of the wavefront with an instance, the remaining steps
            // run without dlib's mask and maximum (6 of 17 instructions per variable).  Instances on their way to the cap --
            // what this kernel is mostly given -- are decided at once.
            step(0, std::true_type{});
            const int part = group_or<G>((active && acc >= eps) ? 1 : 0);
            int go = 1;
            if (ballot_b(have && part == 0) == 0ull) {
#pragma unroll
                for (int l = 1; l < L; ++l) step(l, std::false_type{});
            } else {
#pragma unroll
                for (int l = 1; l < L; ++l) step(l, std::true_type{});
                go = group_or<G>((!active || acc < eps) ? 0 : 1);                         // mpc.h:310-311 (a NaN maximum goes on)
            }
            ++wave_iters;
            ++iter;
            const bool ends = go == 0 || iter >= kn.max_iter;                            // mpc.h:271
            if (ballot_b(have && ends) != 0ull) {
                const bool stop = have && go == 0;
                const bool cap = have && !stop && iter >= kn.max_iter;
                if (stop) { publish(u0_prev[0], u0_prev[1], iter - 1); have = false; }
                if (cap) { flags |= 0x2u; publish(u[0], u[1], iter); have = false; }
                const unsigned long long waiting = ballot_b(!have && !exhausted);
                if (__popcll(waiting) >= refill_groups * G || ballot_b(have) == 0ull) return true;
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


// The same for the general model (per-instance A, B, C, Q, R, bounds, x0, per-step targets; one or two inputs; cold start):
// dlib::mpc<2,I,H>'s iteration in GeneralModel's operation order (mpc_model.h).  Here a recurrence step is 8 chain
// operations per pass ((a00 m0 + a01 m1) + s0, (a10 m0 + a11 m1) + s1; p0 + (a00 n0 + a10 n1), p1 + (a01 n0 + a11 n1)),
// and the chain's start -- M = B u at step 0 (mpc.h:275), N = Q .* M at step H-1 (mpc.h:279) -- is taken by select in
// the one lane that owns it (an arbitrary A has no |a| trick: its entries may be negative, zero or non-finite).
// STATE = true: the caller wants the controller state back (all controls and dlib's v: warm-start chains,
// tpc_mpc_rollout) and may hand one in (GeneralArgs::controls, ::v) -- lane_pg_kernel's job: every instance goes through
// this kernel (queue_len = nullptr: the whole batch, g.n entries of `order`), a lane writes its chunk's part of the state.
// SOLO: as lanex_pg_kernel's.
template <typename T, int I, int H, bool STATE = false, bool SOLO = false>
__global__ __launch_bounds__(64, 1) void lanexg_pg_kernel(GeneralArgs g, Knobs kn, const T* __restrict__ recs,
                                                          const uint32_t* __restrict__ order, uint32_t* __restrict__ ticket,
                                                          unsigned long long* __restrict__ stats,
                                                          const uint32_t* __restrict__ queue_len, int refill_groups,
                                                          uint32_t solo_limit = 0u) {
    using P = LanexPlan<H>;
    if constexpr (SOLO) {
        static_assert(!STATE, "the presolve is a cold-start matter");
        asm volatile("" ::: "a159");
        if (__builtin_nontemporal_load(queue_len) > solo_limit) return;
    }
    static_assert(P::built && P::GA * P::L == H, "chunks of five steps: N = 10, 20, 30, 40");
    static_assert(I == 1 || I == 2, "one or two inputs");
    constexpr int L = P::L, G = P::G, GA = P::GA, NG = P::NG, RL = LaneRec<T, H>::kLen;
    const int64_t n_queue = STATE ? g.n : (int64_t)__builtin_nontemporal_load(queue_len);
    if (n_queue <= 0) return;
    if ((int64_t)blockIdx.x * NG >= n_queue) return;

    const int lane = threadIdx.x;
    const int p = lane & (G - 1);
    const int gbase = lane & ~(G - 1);
    const bool first_chunk = p == 0, last_chunk = p == GA - 1;
    const bool active = p < GA;                   // (N = 30: lanes 6 and 7 of a group hold no chunk; they run along on zeros)
    const T eps = (T)kn.eps;

    GeneralModel<T, I> m;
    m.a00 = m.a01 = m.a10 = m.a11 = m.c0 = m.c1 = m.q0 = m.q1 = m.x00 = m.x01 = (T)0;
#pragma unroll
    for (int j = 0; j < I; ++j) { m.b[0][j] = m.b[1][j] = m.r[j] = m.lo_[j] = m.hi_[j] = (T)0; }
    m.targets = nullptr; m.ld = 0;
    T u[I * L], v[I * L], v2[I * L], mm[I * L];   // (index I * l + j)
    T u0_prev[STATE ? I * L : I];                 // the controls before the speculative update (STATE: the whole chunk)
    T inv_lambda = (T)0, beta = (T)0;
    int64_t k = 0;
    uint32_t iter = 0;
    bool have = false, exhausted = false;
    uint32_t flags = 0;
    uint32_t wave_iters = 0, refills = 0;
#pragma unroll
    for (int q = 0; q < I * L; ++q) { u[q] = (T)0; v[q] = (T)0; v2[q] = (T)0; mm[q] = (T)0; }
#pragma unroll
    for (int j = 0; j < (STATE ? I * L : I); ++j) u0_prev[j] = (T)0;

    // a = this lane's controls (its chunk; chunk 0 holds step 0), vv = dlib's v of the chunk (STATE)
    auto publish = [&](const T* a, const T* vv, uint32_t it) {
        if (p == 0) {
#pragma unroll
            for (int j = 0; j < I; ++j) ((T*)g.u0)[(int64_t)j * g.ld + k] = a[j];
            if (g.iters) g.iters[k] = (int32_t)it;
        }
        if constexpr (STATE) {   // LaneIO<..., GeneralArgs>::write, a chunk per lane
            if (active) {
                if (g.controls) {
                    T* cp = (T*)g.controls + k;
#pragma unroll
                    for (int l = 0; l < L; ++l)
#pragma unroll
                        for (int j = 0; j < I; ++j) cp[(int64_t)((L * p + l) * I + j) * g.ld] = a[I * l + j];
                }
                if (g.v) {
                    T* vp = (T*)g.v + k;
#pragma unroll
                    for (int l = 0; l < L; ++l)
#pragma unroll
                        for (int j = 0; j < I; ++j) vp[(int64_t)((L * p + l) * I + j) * g.ld] = vv[I * l + j];
                }
            }
        }
    };

#pragma unroll 1
    while (true) {
        const unsigned long long want = ballot_b(!have && !exhausted);
        if (want != 0ull && (__popcll(want) >= refill_groups * G || ballot_b(have) == 0ull)) {
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
                    m.load(g, k);
#pragma unroll
                    for (int l = 0; l < L; ++l)
#pragma unroll
                        for (int j = 0; j < I; ++j) u[I * l + j] = active ? rec[2 * (L * (active ? p : 0) + l) + j] : (T)0;
                    const T lambda = rec[2 * H];
                    const uint64_t meta = load_meta<T>(rec + 2 * H + 1);
                    iter = (uint32_t)meta;
                    if (meta & kMetaNonFinite) flags |= 0x1u;
                    if (meta & kMetaBadModel) flags |= 0x4u;
                    const bool vinit = (meta & kMetaVInit) != 0;   // mpc.h:330-334, else a fresh v = 0
#pragma unroll
                    for (int l = 0; l < L; ++l)
#pragma unroll
                        for (int j = 0; j < I; ++j) {
                            T v_in = (T)0;   // (LaneIO<..., GeneralArgs>::load_v)
                            if constexpr (STATE) { if (g.v && active) v_in = ((const T*)g.v + k)[(int64_t)((L * p + l) * I + j) * g.ld]; }
                            v[I * l + j] = vinit ? u[I * l + j] : v_in;
                        }
                    if ((meta & kMetaStopped) || iter >= kn.max_iter) {
                        if (!(meta & kMetaStopped)) flags |= 0x2u;
                        publish(u, v, iter);
                    } else {
                        // the linear term (mpc.h:258-266; linear_term_fn, mpc_model.h): both chains over the whole horizon,
                        // run by every lane of the group, each keeping its chunk
                        T wq[2 * H];
                        linear_term_fn<T, I, H>(m, [&](int q, T val) { wq[q] = val; }, [&](int q) { return wq[q]; },
                                                [&](int q, T val) {
                                                    const int i = q >> 1, j = q & 1;
                                                    if (p == i / L) mm[I * (i % L) + j] = val;
                                                });
                        inv_lambda = (T)1.0 / lambda;                                     // mpc.h:342
                        const T sq = tsqrt(lambda);
                        beta = (sq - (T)1) / (sq + (T)1);                                 // mpc.h:343
                        have = true;
                    }
                }
            }
        }
        if (ballot_b(have) == 0ull) {
            if (ballot_b(!exhausted) == 0ull) break;
            continue;
        }

        auto iteration = [&](T (&vi)[I * L], T (&vo)[I * L]) -> bool {
            // ---- forward pass: M[i] = A M[i-1] + B u[i]  (mpc.h:275-277)
            T s0[L], s1[L], w0[L], w1[L];
#pragma unroll
            for (int l = 0; l < L; ++l) m.first(s0[l], s1[l], &u[I * l]);
            T o0 = (T)0, o1 = (T)0;
#pragma unroll
            for (int r = 0; r < GA; ++r) {
                const T i0 = group_mov<GroupDpp<G, 1, false>::ctrl>(o0), i1 = group_mov<GroupDpp<G, 1, false>::ctrl>(o1);
                T m0 = (m.a00 * i0 + m.a01 * i1) + s0[0];
                T m1 = (m.a10 * i0 + m.a11 * i1) + s1[0];
                m0 = first_chunk ? s0[0] : m0;                                           // mpc.h:275: M = B u at step 0
                m1 = first_chunk ? s1[0] : m1;
                w0[0] = m0; w1[0] = m1;
#pragma unroll
                for (int l = 1; l < L; ++l) {
                    const T t0 = (m.a00 * m0 + m.a01 * m1) + s0[l];
                    const T t1 = (m.a10 * m0 + m.a11 * m1) + s1[l];
                    m0 = t0; m1 = t1;
                    w0[l] = m0; w1[l] = m1;
                }
                o0 = m0; o1 = m1;
            }
            // ---- backward pass: N = Q .* M[i] + trans(A) N  (mpc.h:278-281)
            T p0[L], p1[L], nl0[L], nl1[L];
#pragma unroll
            for (int l = 0; l < L; ++l) { p0[l] = w0[l] * m.q0; p1[l] = w1[l] * m.q1; }
            T b0 = (T)0, b1 = (T)0;
#pragma unroll
            for (int r = 0; r < GA; ++r) {
                const T i0 = group_mov<GroupDpp<G, 1, true>::ctrl>(b0), i1 = group_mov<GroupDpp<G, 1, true>::ctrl>(b1);
                T n0 = p0[L - 1] + (m.a00 * i0 + m.a10 * i1);
                T n1 = p1[L - 1] + (m.a01 * i0 + m.a11 * i1);
                n0 = last_chunk ? p0[L - 1] : n0;                                        // mpc.h:279 at step H-1
                n1 = last_chunk ? p1[L - 1] : n1;
                nl0[L - 1] = n0; nl1[L - 1] = n1;
#pragma unroll
                for (int l = L - 2; l >= 0; --l) {
                    const T t0 = p0[l] + (m.a00 * n0 + m.a10 * n1);
                    const T t1 = p1[l] + (m.a01 * n0 + m.a11 * n1);
                    n0 = t0; n1 = t1;
                    nl0[l] = n0; nl1[l] = n1;
                }
                b0 = n0; b1 = n1;
            }
            // ---- gradient (mpc.h:283), dlib's mask and maximum (mpc.h:298-309), the speculative update (mpc.h:342-343)
#pragma unroll
            for (int j = 0; j < (STATE ? I * L : I); ++j) u0_prev[j] = u[j];
            T acc = (T)0;
            auto step = [&](int l, auto test_c) {
                constexpr bool TEST = decltype(test_c)::value;
#pragma unroll
                for (int j = 0; j < I; ++j) {
                    const int q = I * l + j;
                    const T uu = u[q];
                    const T dd = (mm[q] + m.btm(j, nl0[l], nl1[l])) + uu * m.R(j);        // mpc.h:283
                    if constexpr (TEST) {
                        const T up = (uu <= m.lo(j)) ? (T)0 : dd;                         // mpc.h:298-299
                        const T dn = (uu >= m.hi(j)) ? (T)0 : -dd;
                        acc = tmax(acc, tmax(up, dn));
                    }
                    const T vn = clamp3(uu - inv_lambda * dd, m.lo(j), m.hi(j));          // mpc.h:342
                    u[q] = clamp3(vn + beta * (vn - vi[q]), m.lo(j), m.hi(j));            // mpc.h:343
                    vo[q] = vn;
                }
            };
            // (the stop test decided early: see lanex_pg_kernel)
            step(0, std::true_type{});
            const int part = group_or<G>((active && acc >= eps) ? 1 : 0);
            int go = 1;
            if (ballot_b(have && part == 0) == 0ull) {
#pragma unroll
                for (int l = 1; l < L; ++l) step(l, std::false_type{});
            } else {
#pragma unroll
                for (int l = 1; l < L; ++l) step(l, std::true_type{});
                go = group_or<G>((!active || acc < eps) ? 0 : 1);                         // mpc.h:310-311
            }
            ++wave_iters;
            ++iter;
            const bool ends = go == 0 || iter >= kn.max_iter;                            // mpc.h:271
            if (ballot_b(have && ends) != 0ull) {
                const bool stop = have && go == 0;
                const bool cap = have && !stop && iter >= kn.max_iter;
                if (stop) { publish(u0_prev, vi, iter - 1); have = false; }   // (dlib returns before the update: mpc.h:310-311)
                if (cap) { flags |= 0x2u; publish(u, vo, iter); have = false; }
                const unsigned long long waiting = ballot_b(!have && !exhausted);
                if (__popcll(waiting) >= refill_groups * G || ballot_b(have) == 0ull) return true;
            }
            return false;
        };
#pragma unroll 1
        do {
            if (iteration(v, v2)) {
#pragma unroll
                for (int q = 0; q < I * L; ++q) v[q] = v2[q];
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
