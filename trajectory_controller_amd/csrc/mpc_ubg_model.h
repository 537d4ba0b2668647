// Arithmetic of the LANE_FMA family for the GENERAL model (dlib::mpc<2,I,H> with per-instance A, B, C, Q, R,
// bounds, x0 and per-step targets: reference dlib_files/dlib/control/mpc.h:51-125, :142-163, :253-347), shared by
// the gfx950 kernels (mpc_ubg.h) and the CPU model the tests hold them to bit for bit (tests/model/).
//
// Same contract as mpc_ub_model.h: dlib's iteration and decisions on quantities that differ from dlib's by rounding.
// What carries over from the compact form: fused multiply-adds throughout, the gradient scaled by g = 2^-600 in the
// projected-gradient phase so that the stop test reads "blocked" off the projected step (min(|g df|, |u - u_new|)),
// the momentum step in dlib's difference form.  What does not: per-instance bounds may be degenerate (upper == lower is
// legal, mpc_abstract.h:90-97) or infinite, so the controls stay in dlib's own coordinates (clamp = min + max, or one
// v_med3_f32 in fp32) instead of the unit box; and per-step targets, a free x0 and C make the linear term a per-step
// constant, so it is kept (scaled by g) as dlib keeps it (mpc.h:258-266) instead of being folded into the recurrence.
//   forward   M[i] = A M[i-1] + B u[i]                                   8 fma per step (I = 2)
//   backward  N[i] = g Q .* M[i] + trans(A) N[i+1]                        6
//   gradient  g df[i](j) = fma(B(0,j), N0, fma(B(1,j), N1, fma(g R(j), u(j), g MM[i](j))))     3 per variable
// 40 instructions per horizon step at I = 2 against the bit-exact family's 62.
#pragma once

#include "mpc_ub_model.h"

namespace tpc {
namespace ubg {

using ub::abs_;
using ub::fma_;
// gradient scale of the projected-gradient phase: the compact form's in fp64; 1 in fp32 (its stop test keeps dlib's mask
// as arithmetic, mpc_ubg.h -- per-instance bounds may be pinned or infinite, so there is no batch-wide rounding screen)
template <typename T> struct GradScale : ub::GradScale<T> {};
template <> struct GradScale<float> {
    static constexpr float g = 1.0f, inv_g = 1.0f;
};
using ub::max_;
using ub::min_;
using ub::sqrt_;

template <typename T, int I_> struct Gen {
    static constexpr int I = I_;
    T a00, a01, a10, a11;
    T b[2][I_];
    T c0, c1, q0, q1, gq0, gq1;
    T r[I_], gr[I_], lo[I_], hi[I_];
    T x00, x01;

    TPC_HD void set_scale(T g) {
        gq0 = g * q0; gq1 = g * q1;
        TPC_UNROLL for (int j = 0; j < I_; ++j) gr[j] = g * r[j];
    }
    // dlib's requires clause (mpc_abstract.h:90-97): min(Q) >= 0, min(R) > 0, upper >= lower
    TPC_HD bool invalid() const {
        bool ok = q0 >= (T)0 && q1 >= (T)0;
        TPC_UNROLL for (int j = 0; j < I_; ++j) ok = ok && r[j] > (T)0 && hi[j] >= lo[j];
        return !ok;
    }
    TPC_HD bool nonfinite() const {
        const T big = sizeof(T) == 8 ? (T)1.7976931348623157e308 : (T)3.4028234663852886e38;
        auto fin = [&](T x) { return abs_(x) <= big; };
        bool f = fin(a00) && fin(a01) && fin(a10) && fin(a11) && fin(c0) && fin(c1) && fin(q0) && fin(q1) && fin(x00) && fin(x01);
        TPC_UNROLL for (int j = 0; j < I_; ++j) f = f && fin(b[0][j]) && fin(b[1][j]) && fin(r[j]) && lo[j] == lo[j] && hi[j] == hi[j];
        return !f;
    }
    // dlib's three-argument clamp (matrix_utilities.h:2835-2846); NONAN: one v_med3_f32 on the GPU, same value
    template <bool NONAN = false> TPC_HD T project(T val, int j) const {
#if defined(__HIP_DEVICE_COMPILE__)
        if (NONAN && sizeof(T) == 4) return (T)__builtin_amdgcn_fmed3f((float)val, (float)lo[j], (float)hi[j]);
#endif
        return max_(min_(val, hi[j]), lo[j]);
    }
    // B u
    TPC_HD void bu(T& s0, T& s1, const T* u) const {
        if (I_ == 2) { s0 = fma_(b[0][0], u[0], b[0][I_ - 1] * u[I_ - 1]); s1 = fma_(b[1][0], u[0], b[1][I_ - 1] * u[I_ - 1]); }
        else { s0 = b[0][0] * u[0]; s1 = b[1][0] * u[0]; }
    }
    // forward (mpc.h:275-277): M <- B u (first step), M <- A M + B u
    TPC_HD void first(T& m0, T& m1, const T* u) const { bu(m0, m1, u); }
    TPC_HD void fwd(T& m0, T& m1, const T* u) const {
        T s0, s1;
        bu(s0, s1, u);
        const T n0 = fma_(a00, m0, fma_(a01, m1, s0));
        const T n1 = fma_(a10, m0, fma_(a11, m1, s1));
        m0 = n0; m1 = n1;
    }
    // backward (mpc.h:278-281), scaled by g
    TPC_HD void bwd_last(T& n0, T& n1, T m0, T m1) const { n0 = gq0 * m0; n1 = gq1 * m1; }
    TPC_HD void bwd(T& n0, T& n1, T w0, T w1) const {
        const T t0 = fma_(gq0, w0, fma_(a00, n0, a10 * n1));
        const T t1 = fma_(gq1, w1, fma_(a01, n0, a11 * n1));
        n0 = t0; n1 = t1;
    }
    // g df[i](j) (mpc.h:283); gmm = g MM[i](j)
    TPC_HD T df(int j, T n0, T n1, T u, T gmm) const { return fma_(b[0][j], n0, fma_(b[1][j], n1, fma_(gr[j], u, gmm))); }
};

// g MM = g trans(K) Q (M0 - target) (mpc.h:258-266); target(i, s) returns target[i](s); wput / wget park the 2H
// intermediates; emit(i, j, value) receives g MM[i](j).
template <typename T, int I, int H, class Target, class WPut, class WGet, class Emit>
TPC_HD void linear_term(const Gen<T, I>& m, Target target, WPut wput, WGet wget, Emit emit) {
    T m0 = fma_(m.a00, m.x00, fma_(m.a01, m.x01, m.c0));
    T m1 = fma_(m.a10, m.x00, fma_(m.a11, m.x01, m.c1));
    TPC_UNROLL for (int i = 0; i < H; ++i) {
        if (i > 0) {
            const T n0 = fma_(m.a00, m0, fma_(m.a01, m1, m.c0));
            const T n1 = fma_(m.a10, m0, fma_(m.a11, m1, m.c1));
            m0 = n0; m1 = n1;
        }
        wput(2 * i, m.gq0 * (m0 - target(i, 0)));
        wput(2 * i + 1, m.gq1 * (m1 - target(i, 1)));
    }
    T n0 = wget(2 * (H - 1)), n1 = wget(2 * (H - 1) + 1);
    TPC_UNROLL for (int i = H - 1; i >= 0; --i) {
        if (i < H - 1) {
            const T t0 = wget(2 * i) + fma_(m.a00, n0, m.a10 * n1);
            const T t1 = wget(2 * i + 1) + fma_(m.a01, n0, m.a11 * n1);
            n0 = t0; n1 = t1;
        }
        TPC_UNROLL for (int j = 0; j < I; ++j) emit(i, j, fma_(m.b[0][j], n0, m.b[1][j] * n1));
    }
}

// dlib's constructor quantities (mpc.h:116-123) in dlib's own operation order (no fused operations)
template <typename T, int I, int H, class Emit> TPC_HD T ctor_lambda_qdiag(const Gen<T, I>& m, Emit emit) {
    T sumR = m.r[0];
    if (I == 2) sumR = sumR + m.r[I - 1];
    T lambda = sumR * (T)H;
    T t00 = m.q0, t01 = (T)0, t10 = (T)0, t11 = m.q1;
    TPC_NOUNROLL for (int cidx = 0; cidx < H; ++cidx) {
        T tr = (T)0;
        TPC_UNROLL for (int j = 0; j < I; ++j) {
            const T w0 = m.b[0][j] * t00 + m.b[1][j] * t10;
            const T w1 = m.b[0][j] * t01 + m.b[1][j] * t11;
            const T p = w0 * m.b[0][j] + w1 * m.b[1][j];
            emit(H - cidx - 1, j, p);
            tr = (j == 0) ? p : tr + p;
        }
        lambda = lambda + tr;
        const T u00 = m.a00 * t00 + m.a10 * t10, u01 = m.a00 * t01 + m.a10 * t11;
        const T u10 = m.a01 * t00 + m.a11 * t10, u11 = m.a01 * t01 + m.a11 * t11;
        const T n00 = (u00 * m.a00 + u01 * m.a10) + m.q0;
        const T n01 = (u00 * m.a01 + u01 * m.a11) + (T)0;
        const T n10 = (u10 * m.a00 + u11 * m.a10) + (T)0;
        const T n11 = (u10 * m.a01 + u11 * m.a11) + m.q1;
        t00 = n00; t01 = n01; t10 = n10; t11 = n11;
    }
    return lambda;
}

// Screen of the fast stop test (one failing instance sends the batch through the exact-mask build), as the bit-exact
// family's for this model (mpc_model.h, GeneralModel::fast_stop_ok): with al = max(1, |A|_inf, |A|_1), be = max(|B|_inf,
// |B|_1), U = max |bound|, every intermediate of an iteration is below max|MM| + q be^2 U H^2 al^2H + r U; if that is
// < 1e300 (1e30 in fp32) nothing overflows and no gradient component can be NaN.  The start point 0 must lie strictly
// inside the box, and (fp64) a step that vanishes in rounding must imply |df| < eps: lambda U 2^-50 < eps.
template <typename T, int I> TPC_HD bool fast_stop_ok(const Gen<T, I>& m, T mm_max, T eps, T lambda, int H) {
    constexpr bool D = sizeof(T) == 8;
    const T al = max_((T)1, max_(max_(abs_(m.a00) + abs_(m.a01), abs_(m.a10) + abs_(m.a11)),
                                 max_(abs_(m.a00) + abs_(m.a10), abs_(m.a01) + abs_(m.a11))));
    T be = (T)0, U = (T)0, r_ = (T)0, row0 = (T)0, row1 = (T)0;
    bool straddle = true;
    const T kBmin = (T)(D ? 1e-100 : 1e-10);
    TPC_UNROLL for (int j = 0; j < I; ++j) {
        be = max_(be, abs_(m.b[0][j]) + abs_(m.b[1][j]));
        row0 = row0 + abs_(m.b[0][j]); row1 = row1 + abs_(m.b[1][j]);
        U = max_(U, max_(abs_(m.lo[j]), abs_(m.hi[j])));
        r_ = max_(r_, abs_(m.r[j]));
        straddle = straddle && m.lo[j] <= -kBmin && m.hi[j] >= kBmin;
    }
    be = max_(be, max_(row0, row1));
    T alp = (T)1;
    for (int i = 0; i < 2 * H; ++i) alp = alp * al;
    const T q = max_(abs_(m.q0), abs_(m.q1)), hh = (T)H * (T)H;
    const T bound = mm_max + q * be * be * U * hh * alp + r_ * U;
    bool ok = straddle && bound < (T)(D ? 1e300 : 1e30) && U <= (T)(D ? 1e10 : 1e3) && eps <= (T)(D ? 1e30 : 1e10) &&
              eps >= (T)(D ? 1e-60 : 1e-10);
    if (D) ok = ok && lambda * U * (T)0x1p-50 < eps;
    else ok = ok && lambda <= (T)1e30;
    return ok;
}

}  // namespace ubg
}  // namespace tpc
