// Scalar driver of the LANE_FMA arithmetic (mpc_ub_model.h) for ONE instance of the compact form on the host.
//
// What it computes is what the gfx950 kernels of that family compute (mpc_ub.h: ub_cd_kernel + ub_pg_kernel), the
// same IEEE operations in the same order -- dlib::mpc<2,2,H>::operator() as mpcControllerTobi drives it
// (reference: dlib_files/dlib/control/mpc.h:216-347, src/trajectory_point_follower.cpp:301-389): cold start, <= smo_iters
// coordinate-descent steps, accelerated projected gradient until the largest free gradient component is < eps.
// Two users: tpc_mpc_host.cpp, the product's host path of tpc_mpc_solve_one (a handle created without a device, or
// TPC_MPC_OPT_HOST_SOLVE_ONE), and tests/model/, which holds the kernels to it bit for bit.  Compile with
// -mfma -ffp-contract=off (fma() must be the instruction, and only where the source says so).
#pragma once

#include "mpc_ub_model.h"

namespace tpc {
namespace ub {

// fast_stop_in: the stop test, as ub_pg_kernel's MODE: 0 = dlib's mask by compare and select, 1 = the mask as arithmetic
// (fp32; in fp64 the same as 2), 2 = read off the projected step (what the kernels run where ub::fast_stop_ok -- and, in
// fp32, ub::moved_stop_ok -- hold), -1 = decided here by those screens on this instance (the kernels decide per batch).
// *flags is OR-ed with TPC_MPC_FLAG_NONFINITE (1) / TPC_MPC_FLAG_MAX_ITER (2).
template <typename T, int H, bool EQB>
void host_solve_compact(T v, T ty, T tphi, const T* q, const T* r, double step, double wheelbase, const T* lo,
               const T* hi, double eps_d, unsigned long max_iter, unsigned long smo_iters, int fast_stop_in,
               T* front, T* rear, int* iters, unsigned* flags) {
    const T eps = (T)eps_d;
    Unit<T, EQB> m;
    m.set_uniform((T)1, q, r, lo, hi);
    m.set_instance((T)step, (T)wheelbase, v, ty, tphi);
    T x[2 * H], vv[2 * H], wz[H], wy[H], dd[2 * H], iqd[2 * H];
    for (int i = 0; i < H; ++i) { x[2 * i] = m.xz0; x[2 * i + 1] = m.xz1; }
    const bool nonfinite = m.nonfinite_inputs(ty, tphi);
    const T lambda = ctor_lambda_qdiag<T, H>(m.a, m.c, q[0], q[1], r[0], r[1], [&](int i, int j, T val) {
        iqd[2 * i + j] = val != (T)0 ? (T)1 / (val * m.s(j)) : (T)0;
    });
    const bool fast_stop = fast_stop_in < 0 ? fast_stop_ok(m, ty, tphi, q[0], q[1], r[0], r[1], eps, lambda) : fast_stop_in != 0;
    const bool moved = fast_stop && (sizeof(T) == 8 || (fast_stop_in < 0 ? moved_stop_ok(m, eps, lambda) : fast_stop_in == 2));
    unsigned long iter = 0;
    bool stopped = nonfinite, vinit = false;
    unsigned f = nonfinite ? 1u : 0u;
    // ---- coordinate descent (mpc.h:319-335)
    const unsigned long cd_iters = smo_iters < max_iter ? smo_iters : max_iter;
    for (unsigned long it = 0; it < cd_iters && !stopped; ++it) {
        constexpr bool RVC = Reverse<T, H>::value;   // the forward pass regenerated in the backward sweep
        T Z, Y;
        m.fwd_init(Z, Y);
        for (int i = 0; i < H; ++i) { m.fwd(Z, Y, x[2 * i], x[2 * i + 1]); wz[i] = Z; wy[i] = Y; }
        T n0, n1;
        m.bwd_last(n0, n1, Z, Y);
        for (int i = H - 1; i >= 0; --i) {
            if (i < H - 1) m.bwd(n0, n1, RVC ? Z : wz[i], RVC ? Y : wy[i]);
            dd[2 * i] = m.df0(n1, x[2 * i]);
            dd[2 * i + 1] = m.df1(n0, n1, x[2 * i + 1]);
            if (RVC && i > 0) m.rev(Z, Y, x[2 * i], x[2 * i + 1]);
        }
        T max_df = (T)0;
        int best = 0;
        for (int qv = 0; qv < 2 * H; ++qv) {       // mpc.h:289-309: i then j, strict '>'
            const T up = (x[qv] <= m.bl(qv & 1)) ? (T)0 : dd[qv];
            const T dn = (x[qv] >= m.bh(qv & 1)) ? (T)0 : -dd[qv];
            const T mag = max_(up, dn);
            if (mag > max_df) { max_df = mag; best = qv; }
        }
        if (max_df < eps) { stopped = true; break; }
        if (iqd[best] != (T)0) {
            x[best] = m.project(fma_(-iqd[best], dd[best], x[best]), best & 1);
            vinit = (it + 1 == smo_iters);
        }
        ++iter;
    }
    bool finished = stopped || iter >= max_iter;
    if (finished && !stopped) f |= 2u;
    // ---- accelerated projected gradient (mpc.h:336-345)
    if (!finished) {
        const T g = GradScale<T>::g;
        m.set_uniform(g, q, r, lo, hi);
        m.set_instance((T)step, (T)wheelbase, v, ty, tphi);
        const T geps = g * eps;
        T il0, il1, beta;
        pg_constants<T>(lambda, m.s0, m.s1, il0, il1, beta);
        for (int i = 0; i < H; ++i) {
            vv[2 * i] = vinit ? x[2 * i] : m.xz0;
            vv[2 * i + 1] = vinit ? x[2 * i + 1] : m.xz1;
        }
        const T huge = (T)0x1p100;   // fp32 mask-as-arithmetic stop test (two-fma form)
        while (true) {
            constexpr bool RV = Reverse<T, H>::value;
            T Z, Y;
            m.fwd_init(Z, Y);
            for (int i = 0; i < H; ++i) { m.fwd(Z, Y, x[2 * i], x[2 * i + 1]); wz[i] = Z; wy[i] = Y; }
            const T p0 = x[0], p1 = x[1];
            T n0, n1, acc = (T)0;
            m.bwd_last(n0, n1, Z, Y);
            for (int i = H - 1; i >= 0; --i) {
                if (i < H - 1) m.bwd(n0, n1, RV ? Z : wz[i], RV ? Y : wy[i]);   // RV: (Z, Y) hold step i
                const T xo0 = x[2 * i], xo1 = x[2 * i + 1];
                for (int j = 0; j < 2; ++j) {
                    const int qv = 2 * i + j;
                    const T xx = x[qv];
                    const T d = j == 0 ? m.df0(n1, xx) : m.df1(n0, n1, xx);
                    const T xn = fast_stop ? pg_update<true>(m, j, xx, d, j == 0 ? il0 : il1, beta, vv[qv])
                                           : pg_update<false>(m, j, xx, d, j == 0 ? il0 : il1, beta, vv[qv]);
                    const T vn = vv[qv];
                    T mag;
                    if (!fast_stop) {
                        const T up = (xx <= m.bl(j)) ? (T)0 : d;
                        const T dn = (xx >= m.bh(j)) ? (T)0 : -d;
                        mag = max_(up, dn);
                    } else if (moved) {
                        mag = min_(abs_(d), abs_(xx - vn));
                    } else {
                        const T g_lo = m.gap_lo(j, xx, huge), g_hi = m.gap_hi(j, xx, huge);
                        mag = abs_(max_(min_(d, g_lo), -g_hi));
                    }
                    acc = max_(acc, mag);
                    x[qv] = xn;
                }
                if (RV && i > 0) m.rev(Z, Y, xo0, xo1);   // step i-1 from step i and the controls it was made from
            }
            if (acc < geps) { x[0] = p0; x[1] = p1; break; }   // stop: the controls before this update
            ++iter;
            if (iter >= max_iter) { f |= 2u; break; }
        }
    }
    if (nonfinite) { *front = (T)0; *rear = (T)0; }
    else { *front = m.control(0, x[0]); *rear = m.control(1, x[1]); }
    if (iters) *iters = (int)iter;
    *flags |= f;
}

}  // namespace ub
}  // namespace tpc
