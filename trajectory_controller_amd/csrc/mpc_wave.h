// WAVE kernel: one 64-lane wavefront per MPC instance (the layout BASELINE.json's north_star
// describes).  Lane q owns decision variable q = i*I + j (i = horizon step, j = input).
//
//   prologue  every lane runs dlib's O(H) gradient recurrence on the unit vector e_q: that is
//             column q of the dense Hessian Hd = K'QK + R, kept in the lane's VGPRs as "its row"
//             (Hd is symmetric).  lambda, Q_diag and the linear term MM are computed redundantly by
//             all lanes; each keeps its own element.
//   loop      controls are exchanged through a 2H-entry LDS vector (one ds_write per lane, then
//             broadcast ds_reads); df_q = Hd[q,:].u + MM_q is a register dot product.
//             Coordinate-descent iterations (iter < smo_iters) need the arg-max: DPP wavefront
//             max + ballot, lowest index wins like dlib's strict '>' scan (mpc.h:292-308).
//             Projected-gradient iterations need only "is any free |df| >= eps": one compare and a
//             ballot, no reduction at all.
//
// Not bit-identical to dlib (the dot product sums in a different order; fused multiply-adds are
// used), but it takes dlib's decisions on dlib's quantities, so iteration counts agree and the
// outputs differ by ~1e-11 relative in fp64 (tests/test_parity_gpu.py).
// Iteration counts are wave-uniform: no divergence, no refill.  Supports I*H <= 64.
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

// value of lane K of the caller's 16-lane row, in every lane of that row (DPP row_newbcast)
template <int K> TPC_DEV double row_bcast(double x) {
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(x), 0x150 + K, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(x), 0x150 + K, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
template <int K> TPC_DEV float row_bcast(float x) {
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), 0x150 + K, 0xf, 0xf, false));
}

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

template <typename T, int I, int H, class Args> struct WaveIO;
template <typename T, int I, int H> struct WaveIO<T, I, H, CompactArgs> {
    static TPC_DEV T init_u(const CompactArgs&, int64_t, int, int) { return (T)0; }
    static TPC_DEV T init_v(const CompactArgs&, int64_t, int, int) { return (T)0; }
    static TPC_DEV void write(const CompactArgs& g, int64_t k, bool active, int qi, int qj, T u, T, uint32_t it) {
        if (active && qi == 0) {
            if (qj == 0) ((T*)g.front)[k] = u; else ((T*)g.rear)[k] = u;
        }
        if (g.iters && threadIdx.x == 0) g.iters[k] = (int32_t)it;
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
    static TPC_DEV void write(const GeneralArgs& g, int64_t k, bool active, int qi, int qj, T u, T v, uint32_t it) {
        if (active) {
            if (qi == 0) ((T*)g.u0)[(int64_t)qj * g.ld + k] = u;
            if (g.controls) ((T*)g.controls)[(int64_t)(qi * I + qj) * g.ld + k] = u;
            if (g.v) ((T*)g.v)[(int64_t)(qi * I + qj) * g.ld + k] = v;
        }
        if (g.iters && threadIdx.x == 0) g.iters[k] = (int32_t)it;
    }
};

// Row `slot` (= column, the Hessian is symmetric) of Hd = K'QK + R: dlib's gradient recurrences
// (mpc.h:275-283) applied to the unit vector e_(qi,qj) with MM = 0.  B*e is a column of B at step qi
// and zero elsewhere, so no control vector is materialised; the arithmetic is the generic one
// (x*1 = x, x + 0 = x exactly).  row[2*i + j] = Hd[(i,j)][(qi,qj)].
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
            const T diag = (active && i == qi && j == qj) ? m.R(j) : (T)0;
            row[2 * i + j] = (m.B(0, j) * n0 + m.B(1, j) * n1) + diag;
        }
    }
}

// One instance solved by the calling wavefront (all 64 lanes must call it together; the workgroup
// must be that one wavefront, because the control exchange uses __syncthreads as its wait).
// s_u: I*H + 2 values, s_w: 2*H values of LDS, both 16-byte aligned.
template <typename T, int I, int H, class Model, class Args>
TPC_DEV void wave_solve(const Args& g, const Knobs& kn, int64_t k, T* s_u, T* s_w) {
    constexpr int N = I * H;
    static_assert(N <= kWave, "WAVE kernel: one variable per lane");
    const int lane = threadIdx.x;
    const bool active = lane < N;
    const int qi = active ? lane / I : 0, qj = active ? lane % I : 0;
    const int slot = 2 * qi + qj;

    Model m;
    m.load(g, k);   // every lane reads the same instance: broadcast loads
    const bool nonfinite = m.nonfinite();
    const bool badmodel = m.invalid();   // dlib's requires clause broken: return the start point

    // ---- prologue: this lane's Hessian row, Q_diag, linear-term element, lambda
    T row[2 * H];
    hessian_row<T, I, H>(m, active, qi, qj, row);
    T my_qd = (T)0, my_g = (T)0;
    const T lambda = ctor_lambda_qdiag<T, I, H>(m, [&](int i, int j, T val) { if (2 * i + j == slot) my_qd = val; });
    // every lane computes the same linear term; its 2H intermediates are identical in all lanes,
    // so they are parked in one small LDS vector instead of 2H registers per lane
    linear_term_fn<T, I, H>(m, [&](int q, T val) { s_w[q] = val; }, [&](int q) { return s_w[q]; },
                            [&](int q, T val) { if (q == slot) my_g = val; });
    const T lo = m.lo(qj), hi = m.hi(qj);
    const T eps = (T)kn.eps;
    // the coordinate step divides by Q_diag (mpc.h:325): one correctly rounded reciprocal per lane
    // here instead of a ~13-instruction dependent division chain in every coordinate-descent
    // iteration (the product differs from the quotient by an ulp at most: within this family's
    // tolerance, like its FMA dot product)
    const T my_rqd = (T)1 / my_qd;
    const T inv_lambda = (T)1.0 / lambda;                 // mpc.h:342
    const T sq = tsqrt(lambda);
    const T beta = (sq - (T)1) / (sq + (T)1);             // mpc.h:343

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

    // df_q = Hd[q,:].u + g_q.  Up to 16 variables all live in the first 16-lane row, and each control
    // reaches the others by a DPP row broadcast (two register moves, no memory, no wait); beyond that
    // the controls go through the LDS vector (single-wave workgroup: the barrier is only a wait).
    auto gradient_of = [&](T uu) -> T {
        T a0 = (T)0, a1 = (T)0, a2 = (T)0, a3 = (T)0;
        if constexpr (N <= 16) {
            static_for_w<N>([&](auto qc) {
                constexpr int q = decltype(qc)::value;
                constexpr int e = 2 * (q / I) + (q % I);
                const T uq = row_bcast<q>(uu);
                if constexpr ((q & 3) == 0) a0 = tfma(row[e], uq, a0);
                else if constexpr ((q & 3) == 1) a1 = tfma(row[e], uq, a1);
                else if constexpr ((q & 3) == 2) a2 = tfma(row[e], uq, a2);
                else a3 = tfma(row[e], uq, a3);
            });
        } else {
            if (active) s_u[lane] = uu;
            __syncthreads();
#pragma unroll
            for (int i = 0; i < H; ++i)
#pragma unroll
                for (int j = 0; j < I; ++j) {
                    const int q = i * I + j;
                    const T uq = s_u[q];
                    if ((q & 3) == 0) a0 = tfma(row[2 * i + j], uq, a0);
                    else if ((q & 3) == 1) a1 = tfma(row[2 * i + j], uq, a1);
                    else if ((q & 3) == 2) a2 = tfma(row[2 * i + j], uq, a2);
                    else a3 = tfma(row[2 * i + j], uq, a3);
                }
        }
        return ((a0 + a1) + (a2 + a3)) + my_g;
    };

    uint32_t iter = 0;
    bool capped = true;
    // Two loops, one per phase, each with a single way out; both kinds of step are computed
    // SPECULATIVELY beside the stop test they do not depend on (one wave's fp64 instructions issue every
    // ~5 cycles when independent and every ~9 when each needs the previous result), and nothing is
    // committed before the test has spoken.
    auto run = [&](auto mask_tag) {
        constexpr bool MASK = decltype(mask_tag)::value;
        auto masked = [&](T uu, T df) -> T {
            if constexpr (MASK) {
                const T g_lo = tfma(uu, kHuge, nlo_h), g_hi = tfma(uu, -kHuge, hi_h);
                const T c = tabs(tmax(tmin(df, g_lo), -g_hi));
                return active ? c : (T)0;
            } else {
                const bool blocked = (uu <= lo && df > (T)0) || (uu >= hi && df < (T)0);   // mpc.h:298-299
                return (active && !blocked) ? tabs(df) : (T)0;
            }
        };
        // ---- coordinate descent on the arg-max (mpc.h:319-335)
        const uint32_t cd_end = kn.smo_iters < kn.max_iter ? kn.smo_iters : kn.max_iter;
#pragma unroll 1
        for (; iter < cd_end; ++iter) {
            const T df = gradient_of(u);
            const T c = masked(u, df);
            T nu = put_in_range(lo, hi, -(df - my_qd * u) * my_rqd);             // mpc.h:325-326, every lane its own
            asm volatile("" : "+v"(nu));   // computed HERE, beside the reduction (the optimiser would sink it)
            const T mx = wave_max<N>(c);
            if (mx < eps) { capped = false; return; }                            // mpc.h:310-311
            const unsigned long long hit = __ballot(c == mx);
            const int best = __ffsll((long long)hit) - 1;                        // lowest index wins
            // mpc.h:322: a zero Q_diag skips the update (the iteration still counts).  Decided per lane
            // from the lane's own Q_diag -- no cross-lane read, no wave-uniform branch on the path.
            const bool upd = lane == best && my_qd != (T)0;
            u = upd ? nu : u;
            if (iter + 1 == kn.smo_iters && __ballot(upd) != 0ull) v = u;        // mpc.h:330-334 (last CD iteration only)
        }
        // ---- accelerated projected gradient (mpc.h:336-345); stop test without a reduction
#pragma unroll 1
        for (; iter < kn.max_iter; ++iter) {
            const T df = gradient_of(u);
            const T c = masked(u, df);
            T v_new = clamp3(u - inv_lambda * df, lo, hi);
            T u_new = clamp3(v_new + beta * (v_new - v), lo, hi);
            asm volatile("" : "+v"(v_new), "+v"(u_new));   // computed HERE, beside the stop test
            if (__ballot(c >= eps) == 0ull) { capped = false; return; }          // mpc.h:310-311
            v = v_new;
            u = u_new;
        }
    };
    if ((Model::kScreen && nonfinite) || badmodel) capped = false;
    else if (mask_ok) run(std::true_type{});
    else run(std::false_type{});
    WaveIO<T, I, H, Args>::write(g, k, active, qi, qj, u, v, iter);
    if (g.flags && lane == 0) {
        uint32_t f = 0;
        if (nonfinite) f |= 0x1u;
        if (badmodel) f |= 0x4u;
        if (capped) f |= 0x2u;
        if (f) atomicOr(g.flags, f);
    }
}

template <typename T, int I, int H, class Model, class Args>
__global__ __launch_bounds__(64) void wave_kernel(Args g, Knobs kn) {
    __shared__ __attribute__((aligned(16))) T s_u[I * H + 2];
    __shared__ __attribute__((aligned(16))) T s_w[2 * H];
    wave_solve<T, I, H, Model, Args>(g, kn, (int64_t)blockIdx.x, s_u, s_w);
}

}  // namespace tpc
