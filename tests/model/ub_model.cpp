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

#include "../../trajectory_controller_amd/csrc/mpc_ub_model.h"

namespace {

using namespace tpc::ub;

template <typename T, int H, bool EQB>
void solve_one(T v, T ty, T tphi, const T* q, const T* r, double step, double wheelbase, const T* lo,
               const T* hi, double eps_d, unsigned long max_iter, unsigned long smo_iters, bool fast_stop,
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
        const T huge = (T)0x1p100;   // fp32 stop test (two-fma form)
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
                    } else if (sizeof(T) == 8) {
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

// the kernels' own choice of stop-test build: the screen of mpc_ub_model.h on one instance
template <typename T, int H, bool EQB>
bool screen_one(T v, T ty, T tphi, const T* q, const T* r, double step, double wheelbase, const T* lo, const T* hi, double eps_d) {
    Unit<T, EQB> m;
    m.set_uniform((T)1, q, r, lo, hi);
    m.set_instance((T)step, (T)wheelbase, v, ty, tphi);
    const T lambda = ctor_lambda_qdiag<T, H>(m.a, m.c, q[0], q[1], r[0], r[1], [](int, int, T) {});
    return fast_stop_ok(m, ty, tphi, q[0], q[1], r[0], r[1], (T)eps_d, lambda);
}
template <typename T, bool EQB>
bool screen_dispatch(int H, T v, T ty, T tphi, const T* q, const T* r, double step, double wb, const T* lo, const T* hi, double eps) {
    switch (H) {
#define X(h) case h: return screen_one<T, h, EQB>(v, ty, tphi, q, r, step, wb, lo, hi, eps);
        X(4) X(5) X(10) X(20) X(30) X(40)
#undef X
    }
    return false;
}

template <typename T, bool EQB>
int dispatch(int H, T v, T ty, T tphi, const T* q, const T* r, double step, double wb, const T* lo, const T* hi,
             double eps, unsigned long mi, unsigned long smo, bool fast, T* fo, T* re, int* it, unsigned* fl) {
    switch (H) {
#define X(h) case h: solve_one<T, h, EQB>(v, ty, tphi, q, r, step, wb, lo, hi, eps, mi, smo, fast, fo, re, it, fl); return 0;
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
    if (fast < 0) {   // auto: what the kernels do -- one instance outside the screen sends the batch to the exact build
        fast = 1;
        for (long k = 0; k < n && fast; ++k)
            fast = (eqb ? screen_dispatch<T, true>(H, v[k], dy[k], dphi[k], q, r, step, wb, lo, hi, eps)
                        : screen_dispatch<T, false>(H, v[k], dy[k], dphi[k], q, r, step, wb, lo, hi, eps)) ? 1 : 0;
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
                                                       fast != 0, front + k, rear + k, iters ? iters + k : nullptr, &f)
                                  : dispatch<T, false>(H, v[k], dy[k], dphi[k], q, r, step, wb, lo, hi, eps, mi, smo,
                                                        fast != 0, front + k, rear + k, iters ? iters + k : nullptr, &f);
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

}  // namespace

extern "C" {
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
