// tests/model/ub_model.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// CPU model of the LANE_FMA ("unit-box") kernel family: the arithmetic of
// trajectory_controller_amd/csrc/mpc_ub_model.h driven by a plain scalar loop, one instance at a
// time.  It executes the same IEEE operations in the same order as the gfx950 kernels of that family
// (mpc_ub.h), so the -m gpu tests compare the kernels with it BIT FOR BIT (a kernel bug shows as a
// bit difference), while the model itself is compared with the pinned oracle / the real-dlib fixtures
// under a tolerance on the CPU (-m "not gpu").  It is not a restatement of the reference -- that is
// oracle/ -- and nothing in the product loads it.
//
// Build: tests/model/Makefile (g++ -O2 -mfma -ffp-contract=off).
#include <atomic>
#include <thread>
#include <vector>

#include "../../trajectory_controller_amd/csrc/mpc_ub_host.h"   // the scalar driver of the compact form (shared with the product's host path)
#include "../../trajectory_controller_amd/csrc/mpc_ubg_model.h"

namespace {

using namespace tpc::ub;

// the kernels' own choice of stop-test build: the screens of mpc_ub_model.h on one instance (ub_pg_kernel's MODE)
template <typename T, int H, bool EQB>
int screen_one(T v, T ty, T tphi, const T* q, const T* r, double step, double wheelbase, const T* lo, const T* hi, double eps_d) {
    Unit<T, EQB> m;
    m.set_uniform((T)1, q, r, lo, hi);
    m.set_instance((T)step, (T)wheelbase, v, ty, tphi);
    const T lambda = ctor_lambda_qdiag<T, H>(m.a, m.c, q[0], q[1], r[0], r[1], [](int, int, T) {});
    if (!fast_stop_ok(m, ty, tphi, q[0], q[1], r[0], r[1], (T)eps_d, lambda)) return 0;
    return moved_stop_ok(m, (T)eps_d, lambda) ? 2 : 1;
}
template <typename T, bool EQB>
int screen_dispatch(int H, T v, T ty, T tphi, const T* q, const T* r, double step, double wb, const T* lo, const T* hi, double eps) {
    switch (H) {
#define X(h) case h: return screen_one<T, h, EQB>(v, ty, tphi, q, r, step, wb, lo, hi, eps);
        X(4) X(5) X(10) X(20) X(30) X(40)
#undef X
    }
    return 0;
}

template <typename T, bool EQB>
int dispatch(int H, T v, T ty, T tphi, const T* q, const T* r, double step, double wb, const T* lo, const T* hi,
             double eps, unsigned long mi, unsigned long smo, int fast, T* fo, T* re, int* it, unsigned* fl) {
    switch (H) {
#define X(h) case h: host_solve_compact<T, h, EQB>(v, ty, tphi, q, r, step, wb, lo, hi, eps, mi, smo, fast, fo, re, it, fl); return 0;
        X(4) X(5) X(10) X(20) X(30) X(40)
#undef X
    }
    return -1;
}

template <typename T>
int batch(int H, long n, int nthreads, const T* v, const T* dy, const T* dphi, const T* w4, double step,
          double wb, const T* lo, const T* hi, double eps, unsigned long mi, unsigned long smo, int fast,
          T* front, T* rear, int* iters, unsigned* flags_out) {
    const bool eqb = lo[0] == lo[1] && hi[0] == hi[1];
    const T q[2] = {w4[0], w4[1]}, r[2] = {w4[2], w4[3]};
    // The kernels' batch-wide choice (ub_pg_kernel's MODE): one instance outside the first screen sends the batch to the
    // exact build (0); in fp32 one instance outside the second sends it to the mask-as-arithmetic build (1); else 2.
    // fast < 0: both decided here; fast == 1 ("a fast build", as the callers say it): the second decided here.
    if (fast != 0) {
        int mode = 2;
        for (long k = 0; k < n && mode != 0; ++k) {
            const int one = eqb ? screen_dispatch<T, true>(H, v[k], dy[k], dphi[k], q, r, step, wb, lo, hi, eps)
                                : screen_dispatch<T, false>(H, v[k], dy[k], dphi[k], q, r, step, wb, lo, hi, eps);
            if (one == 0) { if (fast < 0) mode = 0; else if (mode > 1) mode = 1; }
            else if (one < mode) mode = one;
        }
        fast = mode;
    }
    std::atomic<long> next(0);
    std::atomic<unsigned> flags(0);
    std::atomic<int> rc(0);
    auto work = [&]() {
        unsigned f = 0;
        for (;;) {
            const long k0 = next.fetch_add(64);
            if (k0 >= n) break;
            for (long k = k0; k < n && k < k0 + 64; ++k) {
                const int e = eqb ? dispatch<T, true>(H, v[k], dy[k], dphi[k], q, r, step, wb, lo, hi, eps, mi, smo,
                                                       fast, front + k, rear + k, iters ? iters + k : nullptr, &f)
                                  : dispatch<T, false>(H, v[k], dy[k], dphi[k], q, r, step, wb, lo, hi, eps, mi, smo,
                                                        fast, front + k, rear + k, iters ? iters + k : nullptr, &f);
                if (e) rc = e;
            }
        }
        flags |= f;
    };
    std::vector<std::thread> th;
    for (int t = 1; t < nthreads; ++t) th.emplace_back(work);
    work();
    for (auto& t : th) t.join();
    if (flags_out) *flags_out = flags.load();
    return rc.load();
}

// ---- general form (mpc_ubg_model.h): one instance, cold start, AoS inputs like the oracle's batch driver
template <typename T, int I, int H>
void general_one(const T* A, const T* B, const T* C, const T* Q, const T* R, const T* lo, const T* hi, const T* x0,
                 const T* targets /*[H][2]*/, double eps_d, unsigned long max_iter, unsigned long smo_iters, int fast,
                 T* u0, int* iters, unsigned* flags) {
    using namespace tpc::ubg;
    const T eps = (T)eps_d;
    Gen<T, I> m;
    m.a00 = A[0]; m.a01 = A[1]; m.a10 = A[2]; m.a11 = A[3];
    for (int r_ = 0; r_ < 2; ++r_) for (int j = 0; j < I; ++j) m.b[r_][j] = B[r_ * I + j];
    m.c0 = C[0]; m.c1 = C[1]; m.q0 = Q[0]; m.q1 = Q[1];
    for (int j = 0; j < I; ++j) { m.r[j] = R[j]; m.lo[j] = lo[j]; m.hi[j] = hi[j]; }
    m.x00 = x0[0]; m.x01 = x0[1];
    m.set_scale((T)1);
    const bool nonfinite = m.nonfinite(), bad = m.invalid();
    T u[I * H], v[I * H], w[2 * H], gmm[I * H], qd[I * H], dd[I * H];
    for (int q = 0; q < I * H; ++q) u[q] = (T)0;
    const T lambda = ctor_lambda_qdiag<T, I, H>(m, [&](int i, int j, T val) { qd[i * I + j] = val; });
    auto tgt = [&](int i, int s) { return targets[2 * i + s]; };
    T mm_max = (T)0;
    bool mm_nan = false;
    linear_term<T, I, H>(m, tgt, [&](int q, T val) { w[q] = val; }, [&](int q) { return w[q]; },
                         [&](int i, int j, T val) { gmm[i * I + j] = val; mm_max = max_(mm_max, abs_(val)); mm_nan = mm_nan || val != val; });
    if (mm_nan) mm_max = (T)INFINITY;
    if (fast < 0) fast = fast_stop_ok(m, mm_max, eps, lambda, H) ? 1 : 0;
    auto gradient = [&]() {
        T m0, m1;
        m.first(m0, m1, &u[0]);
        w[0] = m0; w[1] = m1;
        for (int i = 1; i < H; ++i) { m.fwd(m0, m1, &u[i * I]); w[2 * i] = m0; w[2 * i + 1] = m1; }
        T n0, n1;
        m.bwd_last(n0, n1, m0, m1);
        for (int i = H - 1; i >= 0; --i) {
            if (i < H - 1) m.bwd(n0, n1, w[2 * i], w[2 * i + 1]);
            for (int j = 0; j < I; ++j) dd[i * I + j] = m.df(j, n0, n1, u[i * I + j], gmm[i * I + j]);
        }
    };
    unsigned long iter = 0;
    bool stopped = bad, vinit = false;   // (dlib propagates non-finite values itself: nothing screened, only flagged)
    unsigned f = (nonfinite ? 1u : 0u) | (bad ? 4u : 0u);
    const unsigned long cd_iters = smo_iters < max_iter ? smo_iters : max_iter;
    for (unsigned long it = 0; it < cd_iters && !stopped; ++it) {
        gradient();
        T max_df = (T)0;
        int best = 0;
        for (int q = 0; q < I * H; ++q) {
            const int j = q % I;
            const T up = (u[q] <= m.lo[j]) ? (T)0 : dd[q];
            const T dn = (u[q] >= m.hi[j]) ? (T)0 : -dd[q];
            const T mag = max_(up, dn);
            if (mag > max_df) { max_df = mag; best = q; }
        }
        if (max_df < eps) { stopped = true; break; }
        if (qd[best] != (T)0) {
            u[best] = m.project(fma_(-((T)1 / qd[best]), dd[best], u[best]), best % I);
            vinit = (it + 1 == smo_iters);
        }
        ++iter;
    }
    const bool finished = stopped || iter >= max_iter;
    if (finished && !stopped) f |= 2u;
    if (!finished) {
        const T g = tpc::ubg::GradScale<T>::g;
        m.set_scale(g);
        for (int q = 0; q < I * H; ++q) { gmm[q] = g * gmm[q]; v[q] = vinit ? u[q] : (T)0; }
        const T geps = g * eps;
        const T il = ((T)1 / lambda) * tpc::ubg::GradScale<T>::inv_g;
        const T sq = sqrt_(lambda);
        const T beta = (sq - (T)1) / (sq + (T)1);
        const T huge = (T)0x1p100;
        while (true) {
            gradient();
            const T p0 = u[0], p1 = I == 2 ? u[I - 1] : (T)0;
            T acc = (T)0;
            for (int q = I * H - 1; q >= 0; --q) {
                const int j = q % I;
                const T uu = u[q], d = dd[q];
                const T vold = v[q];
                const T vn = fast ? m.template project<true>(fma_(-il, d, uu), j) : m.template project<false>(fma_(-il, d, uu), j);
                T mag;
                if (!fast) {
                    const T up = (uu <= m.lo[j]) ? (T)0 : d;
                    const T dn = (uu >= m.hi[j]) ? (T)0 : -d;
                    mag = max_(up, dn);
                } else if (sizeof(T) == 8) {
                    mag = min_(abs_(d), abs_(uu - vn));
                } else {
                    const T g_lo = fma_(uu, huge, -(m.lo[j] * huge)), g_hi = fma_(-huge, uu, m.hi[j] * huge);
                    mag = abs_(max_(min_(d, g_lo), -g_hi));
                }
                acc = max_(acc, mag);
                const T un = fma_(beta, vn - vold, vn);
                u[q] = fast ? m.template project<true>(un, j) : m.template project<false>(un, j);
                v[q] = vn;
            }
            if (acc < geps) { u[0] = p0; if (I == 2) u[I - 1] = p1; break; }
            ++iter;
            if (iter >= max_iter) { f |= 2u; break; }
        }
    }
    for (int j = 0; j < I; ++j) u0[j] = u[j];
    if (iters) *iters = (int)iter;
    *flags |= f;
}

template <typename T, int I, int H>
bool screen_general_one(const T* A, const T* B, const T* C, const T* Q, const T* R, const T* lo, const T* hi, const T* x0,
                        const T* targets, double eps_d) {
    using namespace tpc::ubg;
    Gen<T, I> m;
    m.a00 = A[0]; m.a01 = A[1]; m.a10 = A[2]; m.a11 = A[3];
    for (int r_ = 0; r_ < 2; ++r_) for (int j = 0; j < I; ++j) m.b[r_][j] = B[r_ * I + j];
    m.c0 = C[0]; m.c1 = C[1]; m.q0 = Q[0]; m.q1 = Q[1];
    for (int j = 0; j < I; ++j) { m.r[j] = R[j]; m.lo[j] = lo[j]; m.hi[j] = hi[j]; }
    m.x00 = x0[0]; m.x01 = x0[1];
    m.set_scale((T)1);
    T w[2 * H], mm_max = (T)0;
    bool mm_nan = false;
    const T lambda = ctor_lambda_qdiag<T, I, H>(m, [](int, int, T) {});
    linear_term<T, I, H>(m, [&](int i, int s) { return targets[2 * i + s]; }, [&](int q, T val) { w[q] = val; },
                         [&](int q) { return w[q]; },
                         [&](int, int, T val) { mm_max = max_(mm_max, abs_(val)); mm_nan = mm_nan || val != val; });
    if (mm_nan) mm_max = (T)INFINITY;
    return fast_stop_ok(m, mm_max, (T)eps_d, lambda, H);
}
template <typename T>
bool screen_general(int I, int H, const T* A, const T* B, const T* C, const T* Q, const T* R, const T* lo, const T* hi,
                    const T* x0, const T* targets, double eps) {
#define G(ii, hh) if (I == ii && H == hh) return screen_general_one<T, ii, hh>(A, B, C, Q, R, lo, hi, x0, targets, eps);
    G(1, 4) G(1, 5) G(1, 10) G(1, 20) G(2, 4) G(2, 5) G(2, 10) G(2, 20)
#undef G
    return false;
}

template <typename T>
int general_batch(int I, int H, long n, int nthreads, const T* A, const T* B, const T* C, const T* Q, const T* R,
                  const T* lo, const T* hi, const T* x0, const T* targets, double eps, unsigned long mi, unsigned long smo,
                  int fast, T* u0, int* iters, unsigned* flags_out) {
    auto one = [&](long k, int fst, unsigned* f) -> int {
#define G(ii, hh) if (I == ii && H == hh) { general_one<T, ii, hh>(A + 4 * k, B + 2 * ii * k, C + 2 * k, Q + 2 * k, R + ii * k, lo + ii * k, hi + ii * k, x0 + 2 * k, targets + 2L * hh * k, eps, mi, smo, fst, u0 + ii * k, iters ? iters + k : nullptr, f); return 0; }
        G(1, 4) G(1, 5) G(1, 10) G(1, 20) G(2, 4) G(2, 5) G(2, 10) G(2, 20)
#undef G
        return -1;
    };
    if (fast < 0) {   // the kernels' batch-wide choice: every instance must pass the screen
        fast = 1;
        for (long k = 0; k < n && fast; ++k) {
            fast = screen_general<T>(I, H, A + 4 * k, B + 2 * I * k, C + 2 * k, Q + 2 * k, R + I * k, lo + I * k, hi + I * k,
                                     x0 + 2 * k, targets + 2L * H * k, eps) ? 1 : 0;
        }
    }
    std::atomic<long> next(0);
    std::atomic<unsigned> flags(0);
    std::atomic<int> rc(0);
    auto work = [&]() {
        unsigned f = 0;
        for (;;) {
            const long k0 = next.fetch_add(64);
            if (k0 >= n) break;
            for (long k = k0; k < n && k < k0 + 64; ++k)
                if (one(k, fast, &f)) rc = -1;
        }
        flags |= f;
    };
    std::vector<std::thread> th;
    for (int t = 1; t < nthreads; ++t) th.emplace_back(work);
    work();
    for (auto& t : th) t.join();
    if (flags_out) *flags_out = flags.load();
    return rc.load();
}

}  // namespace

extern "C" {
// Same argument meaning as mpc_oracle_solve_general (oracle/mpc_oracle.h), cold start, u0 only.
int ub_model_solve_general_f64(int I, int H, long n, int nthreads, const double* A, const double* B, const double* C,
                               const double* Q, const double* R, const double* lo, const double* hi, const double* x0,
                               const double* targets, double eps, unsigned long max_iter, unsigned long smo_iters,
                               int fast_stop, double* u0, int* iters, unsigned* flags) {
    return general_batch<double>(I, H, n, nthreads, A, B, C, Q, R, lo, hi, x0, targets, eps, max_iter, smo_iters, fast_stop, u0, iters, flags);
}
int ub_model_solve_general_f32(int I, int H, long n, int nthreads, const float* A, const float* B, const float* C,
                               const float* Q, const float* R, const float* lo, const float* hi, const float* x0,
                               const float* targets, double eps, unsigned long max_iter, unsigned long smo_iters,
                               int fast_stop, float* u0, int* iters, unsigned* flags) {
    return general_batch<float>(I, H, n, nthreads, A, B, C, Q, R, lo, hi, x0, targets, eps, max_iter, smo_iters, fast_stop, u0, iters, flags);
}

// Same argument meaning as mpc_oracle_solve_compact (oracle/mpc_oracle.h); fast_stop selects the
// stop-test form of the kernels' screened build (1), dlib's masked form (0), or -1: decided by the kernels' own screen.
int ub_model_solve_compact_f64(int H, long n, int nthreads, const double* v, const double* dy, const double* dphi,
                               const double* weights4, double T, double l, const double* lo2, const double* hi2,
                               double eps, unsigned long max_iter, unsigned long smo_iters, int fast_stop,
                               double* front, double* rear, int* iters, unsigned* flags) {
    return batch<double>(H, n, nthreads, v, dy, dphi, weights4, T, l, lo2, hi2, eps, max_iter, smo_iters, fast_stop,
                         front, rear, iters, flags);
}
int ub_model_solve_compact_f32(int H, long n, int nthreads, const float* v, const float* dy, const float* dphi,
                               const float* weights4, double T, double l, const float* lo2, const float* hi2,
                               double eps, unsigned long max_iter, unsigned long smo_iters, int fast_stop,
                               float* front, float* rear, int* iters, unsigned* flags) {
    return batch<float>(H, n, nthreads, v, dy, dphi, weights4, T, l, lo2, hi2, eps, max_iter, smo_iters, fast_stop,
                        front, rear, iters, flags);
}
}
