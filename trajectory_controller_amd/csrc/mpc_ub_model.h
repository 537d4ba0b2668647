// Arithmetic of the LANE_FMA kernel family ("unit-box" form), shared by the gfx950 kernels
// (mpc_ub.h) and by the CPU model the tests check them against bit for bit (tests/model/).
//
// What the family computes is dlib::mpc<2,2,H>::solve_linear_mpc for the compact model
// (reference: dlib_files/dlib/control/mpc.h:253-347 driven as in
// src/trajectory_point_follower.cpp:326-380) -- the same iteration: <= smo_iters coordinate-descent
// steps on the arg-max free gradient component, then accelerated projected gradient with step
// 1/lambda, stop when the largest free gradient component is < eps -- but NOT in dlib's operation
// order.  It is the tolerance-grade sibling of the bit-exact LANE family: every decision dlib takes
// is taken on the same quantity, the quantities differ from dlib's by rounding (~1e-14 on the
// BASELINE workloads), and an iteration count can differ only where dlib's max|df| comes within
// that rounding of eps.  Three changes buy ~half the instructions of an iteration:
//
//   1. controls in unit-box coordinates   u_j = lo_j + s_j * x_j,  s_j = hi_j - lo_j,  0 <= x_j <= 1.
//      dlib's three-argument clamp (matrix_utilities.h:2835-2846) becomes the [0,1] output clamp
//      that every gfx950 VOP3 instruction carries for free, so both clamps of mpc.h:342-343 cost
//      nothing; "u sits on a bound" (mpc.h:298-299) is x == 0 or x == 1, exactly.
//   2. fused multiply-adds everywhere (the reference binary has none).
//   3. no stored linear term.  dlib precomputes MM = trans(K) Q (M0 - target) (mpc.h:258-266) and
//      adds it to every gradient; by linearity the same gradient comes out of ONE backward pass
//      driven by the predicted state ERROR  E[i] = M[i] - target  (x0 = 0, C = 0 for this model,
//      so dlib's M0[i] is zero), at one extra subtraction per step and no memory at all.
//
// With Z[i] = M[i](0) - ty, Y[i] = M[i](1) + lo1 (the shift absorbs B*lo into the recurrence):
//   forward   Z[i] = fma(a s1, x[i](1), fma(a, Y[i-1], Z[i-1]))            Z[-1] = -ty
//             Y[i] = fma(c s0, x[i](0), fma(-c s1, x[i](1), Y[i-1])) (+ c (lo0 - lo1))   Y[-1] = lo1
//   backward  N0[i] = fma(g q0, Z[i], N0[i+1])
//             N1[i] = fma(a, N0[i+1], N1[i+1]) + fma(g q1, Y[i], -g q1 (lo1 + tphi))
//   gradient  df[i](0) = fma(c, N1[i], fma(g r0 s0, x[i](0), g r0 lo0))
//             df[i](1) = fma(a, N0[i], fma(-c, N1[i], fma(g r1 s1, x[i](1), g r1 lo1)))
// g scales the whole gradient: 1 in the coordinate-descent phase, 2^-600 (fp64) in the projected-
// gradient phase, where it lets the stop test read "blocked" off the projected step with no extra
// multiply: min(|g df|, |x - x_new|) is >= g eps exactly where dlib's masked |df| is >= eps (the
// step of a free variable is |df| / (lambda s) >> g |df|; a blocked variable's is exactly 0).
// 25 instructions per horizon step instead of the bit-exact family's 49.
//
// Everything is written with explicit fma / mul / add calls and compiled with -ffp-contract=off on
// both sides, so the CPU model and the kernels execute the same IEEE operations.
#pragma once

#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define TPC_HD __host__ __device__ __forceinline__
#define TPC_UNROLL _Pragma("unroll")
#define TPC_NOUNROLL _Pragma("unroll 1")
#else
#define TPC_HD inline
#define TPC_UNROLL
#define TPC_NOUNROLL
#endif

namespace tpc {
namespace ub {

TPC_HD double fma_(double a, double b, double c) { return __builtin_fma(a, b, c); }
TPC_HD float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
TPC_HD double abs_(double x) { return __builtin_fabs(x); }
TPC_HD float abs_(float x) { return __builtin_fabsf(x); }
TPC_HD double max_(double a, double b) { return __builtin_fmax(a, b); }
TPC_HD float max_(float a, float b) { return __builtin_fmaxf(a, b); }
TPC_HD double min_(double a, double b) { return __builtin_fmin(a, b); }
TPC_HD float min_(float a, float b) { return __builtin_fminf(a, b); }
TPC_HD double sqrt_(double x) { return __builtin_sqrt(x); }
TPC_HD float sqrt_(float x) { return __builtin_sqrtf(x); }
// [0,1] clamp in the operand order LLVM folds into the producing instruction's clamp bit
template <typename T> TPC_HD T clamp01(T x) { return min_(max_(x, (T)0), (T)1); }

// Gradient scale of the projected-gradient phase and its inverse (powers of two: exact).
template <typename T> struct GradScale;
template <> struct GradScale<double> {
    static constexpr double g = 0x1p-600, inv_g = 0x1p600;
};
// fp32: 2^-40 (round 4; it was 1).  The scale is exact (a power of two; nothing in the projected-gradient phase comes
// near fp32's range at either end), and it lets fp32 use the same arithmetic stop test wherever ub::moved_stop_ok holds.
template <> struct GradScale<float> {
    static constexpr float g = 0x1p-40f, inv_g = 0x1p40f;
};

// Unit-box coordinates are an fp64 device.  In fp32 they would cost accuracy: a control near zero sits at
// x ~ 0.5, where one ulp of x is 6e-8 of the box however small the control is, so steps dlib's float
// arithmetic still resolves vanish here (measured: +10 % iterations at N = 20, +57 % at N = 30, results
// further from the fp64 ones) -- and fp32 has v_med3_f32, dlib's clamp in one instruction, so the box buys
// little there.  fp32 therefore keeps dlib's own coordinates: the same recurrences with the identity map
// (s = 1, offset 0) and the box [lower, upper].
template <typename T> struct UnitBox { static constexpr bool value = sizeof(T) == 8; };

// The compact model in the family's coordinates  u_j = lo_j + s_j x_j,  x_j in [bl_j, bh_j].
// UBOX: the unit box (s = upper - lower, offset lower, box [0, 1]); otherwise dlib's coordinates (s = 1,
// offset 0, box [lower, upper]).  EQB: both inputs share their bounds (the reference's configuration,
// src/trajectory_point_follower.cpp:16-18), which drops one addition per step (always so without UBOX).
template <typename T, bool EQB, bool UBOX = UnitBox<T>::value> struct Unit {
    static constexpr bool kUnitBox = UBOX;
    // uniform over a batch
    T s0, s1, lo0, lo1, hi0, hi1;             // the map (hi: dlib's upper, for exact outputs on the bound)
    T bl0, bl1, bh0, bh1;                     // the box in x coordinates
    T xz0, xz1;                               // u = 0 (dlib's start point, mpc.h:110) in x coordinates
    T gq0, gq1, grs0, grs1, grl0, grl1;       // g q, g r s, g r lo
    // per instance
    T a, c, as1, cs0, cs1, dlt, z0, q1th;

    TPC_HD T s(int j) const { return j == 0 ? s0 : s1; }
    TPC_HD T lo(int j) const { return j == 0 ? lo0 : lo1; }
    TPC_HD T hi(int j) const { return j == 0 ? hi0 : hi1; }
    TPC_HD T xz(int j) const { return j == 0 ? xz0 : xz1; }
    TPC_HD T bl(int j) const { return UBOX ? (T)0 : (j == 0 ? bl0 : bl1); }
    TPC_HD T bh(int j) const { return UBOX ? (T)1 : (j == 0 ? bh0 : bh1); }

    // dlib's three-argument clamp (matrix_utilities.h:2835-2846) onto the box.  NONAN (the screened builds):
    // one v_med3_f32 on the GPU, the same value for every non-NaN operand.
    template <bool NONAN = false> TPC_HD T project(T val, int j) const {
        if (UBOX) return clamp01(val);
#if defined(__HIP_DEVICE_COMPILE__)
        if (NONAN && sizeof(T) == 4) return (T)__builtin_amdgcn_fmed3f((float)val, (float)bl(j), (float)bh(j));
#endif
        return max_(min_(val, bh(j)), bl(j));
    }

    // q, r, lo, hi: dlib's Q, R, lower, upper (src/trajectory_point_follower.cpp:359-363, :16-18)
    TPC_HD void set_uniform(T g, const T* q, const T* r, const T* lo, const T* hi) {
        hi0 = hi[0]; hi1 = hi[1];
        if (UBOX) {
            lo0 = lo[0]; lo1 = lo[1];
            s0 = hi0 - lo0; s1 = hi1 - lo1;
            xz0 = ((T)0 - lo0) / s0; xz1 = ((T)0 - lo1) / s1;
            bl0 = bl1 = (T)0; bh0 = bh1 = (T)1;
        } else {
            lo0 = lo1 = (T)0;
            s0 = s1 = (T)1;
            xz0 = xz1 = (T)0;
            bl0 = lo[0]; bl1 = lo[1]; bh0 = hi[0]; bh1 = hi[1];
        }
        gq0 = g * q[0]; gq1 = g * q[1];
        const T gr0 = g * r[0], gr1 = g * r[1];
        grs0 = gr0 * s0; grs1 = gr1 * s1;
        grl0 = gr0 * lo0; grl1 = gr1 * lo1;
    }
    // a = T v, c = T v / l  (src/trajectory_point_follower.cpp:327-330); target (ty, tphi) (:371)
    TPC_HD void set_instance(T step, T wheelbase, T v, T ty, T tphi) {
        a = step * v;
        c = step * v / wheelbase;
        as1 = a * s1; cs0 = c * s0; cs1 = c * s1;
        dlt = c * (lo0 - lo1);
        z0 = (T)0 - ty;
        q1th = gq1 * (lo1 + tphi);
    }
    // the same from a and c themselves (a refill pass reads them from the instance's record instead of dividing again)
    TPC_HD void set_instance_ac(T a_, T c_, T ty, T tphi) {
        a = a_;
        c = c_;
        as1 = a * s1; cs0 = c * s0; cs1 = c * s1;
        dlt = c * (lo0 - lo1);
        z0 = (T)0 - ty;
        q1th = gq1 * (lo1 + tphi);
    }
    TPC_HD bool nonfinite_inputs(T ty, T tphi) const {
        const T big = sizeof(T) == 8 ? (T)1.7976931348623157e308 : (T)3.4028234663852886e38;
        return !(abs_(a) <= big && abs_(c) <= big && abs_(ty) <= big && abs_(tphi) <= big);
    }
    // forward pass, one step (mpc.h:275-277 in the coordinates above); (Z, Y) <- step i from step i-1
    TPC_HD void fwd_init(T& Z, T& Y) const { Z = z0; Y = lo1; }
    TPC_HD void fwd(T& Z, T& Y, T x0, T x1) const {
        const T zn = fma_(as1, x1, fma_(a, Y, Z));
        T yn = fma_(cs0, x0, fma_(-cs1, x1, Y));
        if (!EQB && UBOX) yn = yn + dlt;
        Z = zn; Y = yn;
    }
    // The forward recurrence run backwards: (Z, Y) of step i-1 from those of step i and x[i].  Used by the
    // plans that do not keep the forward pass (Reverse below): the backward sweep regenerates each step's
    // (Z, Y) from the next one's at four operations per step.  The regenerated values differ from the
    // forward ones by rounding (a few ulp over the horizon), like everything else in this family.
    TPC_HD void rev(T& Z, T& Y, T x0, T x1) const {
        T yp = Y;
        if (!EQB && UBOX) yp = yp - dlt;
        yp = fma_(cs1, x1, fma_(-cs0, x0, yp));
        Z = fma_(-a, yp, fma_(-as1, x1, Z));
        Y = yp;
    }
    // backward pass (mpc.h:278-281): the last step, then one step
    TPC_HD void bwd_last(T& n0, T& n1, T Z, T Y) const {
        n0 = gq0 * Z;
        n1 = fma_(gq1, Y, -q1th);
    }
    TPC_HD void bwd(T& n0, T& n1, T Z, T Y) const {
        const T e1 = fma_(gq1, Y, -q1th);
        const T t1 = fma_(a, n0, n1) + e1;
        n0 = fma_(gq0, Z, n0);
        n1 = t1;
    }
    // gradient components of one step (mpc.h:283)
    TPC_HD T df0(T n1, T x0) const { return fma_(c, n1, fma_(grs0, x0, grl0)); }
    TPC_HD T df1(T n0, T n1, T x1) const { return fma_(a, n0, fma_(-c, n1, fma_(grs1, x1, grl1))); }
    // x -> control; the bounds and the untouched start point come out exactly
    TPC_HD T control(int j, T x) const {
        if (!UBOX) return x;
        return x == (T)1 ? hi(j) : (x == xz(j) ? (T)0 : fma_(s(j), x, lo(j)));
    }
    // fp32 stop test (dlib's mask as arithmetic): (x - bl) 2^100 and (bh - x) 2^100, zero exactly on the bound
    TPC_HD T gap_lo(int j, T x, T huge) const { return fma_(x, huge, -(bl(j) * huge)); }
    TPC_HD T gap_hi(int j, T x, T huge) const { return fma_(-huge, x, bh(j) * huge); }
};

// Screen of the fast stop test (per instance; one failing instance sends the batch through the
// exact build, like LANE).  Needed: no intermediate can overflow or be NaN, the start point u = 0
// lies inside the box, and a projected step that vanishes in rounding implies |df| < eps:
// lambda * s * 2^-50 < eps (fp32: finiteness only here; its own rounding condition is ub::moved_stop_ok below).
// fp64 magnitudes with |a|,|c|,|target| <= 1e50, q <= 1e30, r <= 1e100, s,|bound| <= 1e10, H <= 40:
// |Y| <= 4e61, |Z| <= 2e113, |N0| <= 7e144, |N1| <= 3e196, |df| <= 3e246.
template <typename T, bool EQB>
TPC_HD bool fast_stop_ok(const Unit<T, EQB>& m, T ty, T tphi, T q0, T q1, T r0, T r1, T eps, T lambda) {
    constexpr bool D = sizeof(T) == 8;
    constexpr T kAl = (T)(D ? 1e50 : 1e3), kQ = (T)(D ? 1e30 : 1e3), kR = (T)(D ? 1e100 : 1e10);
    constexpr T kS = (T)(D ? 1e10 : 1e2), kEpsLo = (T)(D ? 1e-60 : 1e-10), kEpsHi = (T)(D ? 1e30 : 1e10);
    const T smax = max_(m.s0, m.s1);
    bool ok = abs_(m.a) <= kAl && abs_(m.c) <= kAl && abs_(ty) <= kAl && abs_(tphi) <= kAl;
    ok = ok && abs_(q0) <= kQ && abs_(q1) <= kQ && abs_(r0) <= kR && abs_(r1) <= kR;
    ok = ok && eps >= kEpsLo && eps <= kEpsHi;
    if (Unit<T, EQB>::kUnitBox) {
        ok = ok && smax <= kS && abs_(m.lo0) <= kS && abs_(m.lo1) <= kS && abs_(m.hi0) <= kS && abs_(m.hi1) <= kS;
        ok = ok && m.xz0 >= (T)0 && m.xz0 <= (T)1 && m.xz1 >= (T)0 && m.xz1 <= (T)1;
        ok = ok && lambda * smax * (T)0x1p-50 < eps;
    } else {
        // dlib's coordinates (fp32): the start point 0 strictly inside the box, and a gap of one ulp off a
        // bound, times 2^100, beyond every admissible eps: |bound| >= 1e-10
        constexpr T kBmin = (T)1e-10;
        ok = ok && m.bl0 <= -kBmin && m.bl1 <= -kBmin && m.bh0 >= kBmin && m.bh1 >= kBmin;
        ok = ok && m.bl0 >= -kS && m.bl1 >= -kS && m.bh0 <= kS && m.bh1 <= kS && lambda <= (T)1e30;
    }
    return ok;
}


// Second screen, fp32 in dlib's coordinates only (the unit-box form has its counterpart inside fast_stop_ok): may the
// stop test be read off the projected step -- min(|g df|, |x - x_new|) -- instead of dlib's mask as arithmetic?
// With B = the largest |bound|, b = the smallest:
//   a free variable with |df| >= eps must MOVE: its step |df| / lambda must exceed half an ulp of x, at most B 2^-24;
//   and by more than g eps: lambda <= 2^36 (the step is then >= 2^-36 eps, rounding takes at most a third of it);
//   a step clipped at a bound from inside still shows: the spacing of floats next to a bound, >= b 2^-24, exceeds g eps;
//   a blocked variable's step is exactly 0 (the clamp returns the bound's own bits).
// lambda grows with the horizon and the speed (N = 20, v = 4 m/s: 3.0e5, the margin is 1.4; N = 40: 4.9e6, refused),
// so batches that fail here keep the mask-as-arithmetic build: same decisions, 14 % slower at N = 20.
template <typename T, bool EQB> TPC_HD bool moved_stop_ok(const Unit<T, EQB>& m, T eps, T lambda) {
    if (Unit<T, EQB>::kUnitBox) return true;
    const T bmax = max_(max_(abs_(m.bl0), abs_(m.bl1)), max_(abs_(m.bh0), abs_(m.bh1)));
    const T bmin = min_(min_(abs_(m.bl0), abs_(m.bl1)), min_(abs_(m.bh0), abs_(m.bh1)));
    bool ok = lambda * bmax * (T)0x1p-24 < eps * (T)0.999 && lambda <= (T)0x1p36;
    ok = ok && bmin * (T)0x1p-24 >= (T)2 * GradScale<T>::g * eps;
    return ok;
}

// Which (dtype, horizon) regenerate the forward pass in the backward sweep instead of keeping it: part of
// the family's arithmetic, so it lives here where kernels and model both see it.
// fp64 at N = 30 and 40: x alone takes 120 / 160 of the 256 registers a VALU instruction can name, and a kept
// (even a checkpointed) forward pass sent ~300 values per iteration through the AGPRs; regenerating it costs 4
// operations per step and no register.  (At N = 20 the kept forward pass fits and is faster: 5.35 against 6.1 ms.)
template <typename T, int H> struct Reverse {
    static constexpr bool value = sizeof(T) == 8 && H >= 30;
};

// dlib's constructor quantities for the compact model (mpc.h:116-123), in dlib's own operation order
// (no fused operations): lambda, and through emit(i, j, value) Q_diag[i](j).
template <typename T, int H, class Emit> TPC_HD T ctor_lambda_qdiag(T a, T c, T q0, T q1, T r0, T r1, Emit emit) {
    // A = [1 a; 0 1], B = [0 a; c -c]
    const T A[2][2] = {{(T)1, a}, {(T)0, (T)1}};
    const T B[2][2] = {{(T)0, a}, {c, (T)0 - c}};
    T lambda = (r0 + r1) * (T)H;
    T t00 = q0, t01 = (T)0, t10 = (T)0, t11 = q1;
    for (int cidx = 0; cidx < H; ++cidx) {
        // W(r,:) = trans(B)(r,:) * T ; P(r,r) = W(r,:) * B(:,r)
        auto diag = [&](T b0, T b1) {
            const T w0 = b0 * t00 + b1 * t10;
            const T w1 = b0 * t01 + b1 * t11;
            return w0 * b0 + w1 * b1;
        };
        const T p0 = diag(B[0][0], B[1][0]), p1 = diag(B[0][1], B[1][1]);
        emit(H - cidx - 1, 0, p0);
        emit(H - cidx - 1, 1, p1);
        const T tr = p0 + p1;
        lambda = lambda + tr;
        const T u00 = A[0][0] * t00 + A[1][0] * t10, u01 = A[0][0] * t01 + A[1][0] * t11;
        const T u10 = A[0][1] * t00 + A[1][1] * t10, u11 = A[0][1] * t01 + A[1][1] * t11;
        const T n00 = (u00 * A[0][0] + u01 * A[1][0]) + q0;
        const T n01 = (u00 * A[0][1] + u01 * A[1][1]) + (T)0;
        const T n10 = (u10 * A[0][0] + u11 * A[1][0]) + (T)0;
        const T n11 = (u10 * A[0][1] + u11 * A[1][1]) + q1;
        t00 = n00; t01 = n01; t10 = n10; t11 = n11;
    }
    return lambda;
}

// Step constants of the projected-gradient phase (mpc.h:342-343) in unit coordinates:
// v_new = clamp01(x - il_j * (g df)),  il_j = 1 / (g lambda s_j);  beta = (sqrt(lambda)-1)/(sqrt(lambda)+1)
template <typename T> TPC_HD void pg_constants(T lambda, T s0, T s1, T& il0, T& il1, T& beta) {
    il0 = ((T)1 / (lambda * s0)) * GradScale<T>::inv_g;
    il1 = ((T)1 / (lambda * s1)) * GradScale<T>::inv_g;
    const T sq = sqrt_(lambda);
    beta = (sq - (T)1) / (sq + (T)1);
}
// One variable's projected-gradient update (mpc.h:342-343): returns the new x, leaves the new v in `v`.
// The momentum term must stay in dlib's difference form v + beta (v - v_old): only that form returns v
// EXACTLY when v == v_old, which is what keeps a variable that sits on a bound on it (the algebraically equal
// fma(1 + beta, v, -beta v_old) leaves the bound by an ulp, and the stop test then sees a free variable).
template <bool NONAN, class M, typename T> TPC_HD T pg_update(const M& m, int j, T x, T d, T il, T beta, T& v) {
    const T vold = v;
    v = m.template project<NONAN>(fma_(-il, d, x), j);
    return m.template project<NONAN>(fma_(beta, v - vold, v), j);
}

}  // namespace ub
}  // namespace tpc
