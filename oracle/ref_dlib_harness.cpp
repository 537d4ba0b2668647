// oracle/ref_dlib_harness.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// A thin C-ABI wrapper around the REAL reference solver, dlib::mpc<S,I,H>
// (reference: dlib_files/dlib/control/mpc.h:18-365), compiled from the headers where they
// lie under /root/reference/dlib_files (passed with -I by oracle/Makefile; nothing from the
// reference tree is copied into this repository).  The built object goes to oracle/_ref/
// (git-ignored).  It is used
//   * by tests/golden/make_golden.py to generate the committed golden vectors,
//   * by tests/ to validate the C restatement (oracle/mpc_oracle.c) bit-for-bit,
//   * by bench.py's cpu_baseline leg (kind "reference") as the timed dlib-CPU path.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
//
// Call patterns replayed:
//   dlibref_solve_compact  : src/trajectory_point_follower.cpp:326-384 (mpcControllerTobi:
//                            model from v, fresh controller, one target, x0 = 0, cold start)
//   dlibref_solve_general  : fresh controller, per-step targets (mpc.h:142-155), any x0, C
//   dlibref_rollout        : one controller object called repeatedly (warm start + target
//                            shift, mpc.h:229-239), the closed loop of dlib/test/mpc.cpp:295-316

#include <dlib/control.h>
#include <dlib/revision.h>

#include <cstdint>
#include <thread>
#include <vector>
#include <algorithm>

namespace {

template <long I>
struct Model {
    dlib::matrix<double, 2, 2> A;
    dlib::matrix<double, 2, I> B;
    dlib::matrix<double, 2, 1> C, Q;
    dlib::matrix<double, I, 1> R, lo, hi;
};

// Row-major flat arrays -> dlib fixed matrices.
template <long I>
Model<I> load_model(const double* A, const double* B, const double* C, const double* Q,
                    const double* R, const double* lo, const double* hi) {
    Model<I> m;
    for (long r = 0; r < 2; ++r)
        for (long c = 0; c < 2; ++c) m.A(r, c) = A[r * 2 + c];
    for (long r = 0; r < 2; ++r)
        for (long c = 0; c < I; ++c) m.B(r, c) = B[r * I + c];
    for (long r = 0; r < 2; ++r) { m.C(r) = C[r]; m.Q(r) = Q[r]; }
    for (long r = 0; r < I; ++r) { m.R(r) = R[r]; m.lo(r) = lo[r]; m.hi(r) = hi[r]; }
    return m;
}

// ---- compact form: the reference module's own use of the solver -------------------------
template <unsigned long H>
void compact_range(long begin, long end, const double* v, const double* dy, const double* dphi,
                   const double* w, double T, double l, const double* lo, const double* hi,
                   double eps, unsigned long max_iter, double* out_front, double* out_rear) {
    for (long k = begin; k < end; ++k) {
        const double vk = v[k];
        dlib::matrix<double, 2, 2> A;
        A = 1, T * vk, 0, 1;
        dlib::matrix<double, 2, 2> B;
        B = 0, T * vk, T * vk / l, -T * vk / l;
        dlib::matrix<double, 2, 1> C;
        C = 0, 0;
        dlib::matrix<double, 2, 1> Q;
        Q = w[0], w[1];
        dlib::matrix<double, 2, 1> R;
        R = w[2], w[3];
        dlib::matrix<double, 2, 1> lower, upper;
        lower = lo[0], lo[1];
        upper = hi[0], hi[1];
        dlib::mpc<2, 2, H> controller(A, B, C, Q, R, lower, upper);
        dlib::matrix<double, 2, 1> target;
        target = dy[k], dphi[k];
        controller.set_target(target);
        controller.set_epsilon(eps);
        controller.set_max_iterations(max_iter);
        dlib::matrix<double, 2, 1> x0;
        x0 = 0, 0;
        dlib::matrix<double, 2, 1> action = controller(x0);
        out_front[k] = action(0, 0);
        out_rear[k] = action(1, 0);
    }
}

template <unsigned long H>
int compact_threads(long n, int nthreads, const double* v, const double* dy, const double* dphi,
                    const double* w, double T, double l, const double* lo, const double* hi,
                    double eps, unsigned long max_iter, double* out_front, double* out_rear) {
    if (nthreads <= 1) {
        compact_range<H>(0, n, v, dy, dphi, w, T, l, lo, hi, eps, max_iter, out_front, out_rear);
        return 0;
    }
    std::vector<std::thread> pool;
    // interleaved blocks of 64 so iteration-count skew spreads over threads
    const long chunk = 64;
    for (int t = 0; t < nthreads; ++t) {
        pool.emplace_back([=]() {
            for (long b = (long)t * chunk; b < n; b += (long)nthreads * chunk)
                compact_range<H>(b, std::min(n, b + chunk), v, dy, dphi, w, T, l, lo, hi, eps,
                                 max_iter, out_front, out_rear);
        });
    }
    for (auto& th : pool) th.join();
    return 0;
}

// ---- general form -----------------------------------------------------------------------
template <long I, unsigned long H>
void general_one(const Model<I>& m, const double* x0, const double* targets, double eps,
                 unsigned long max_iter, double* u0) {
    dlib::mpc<2, I, H> controller(m.A, m.B, m.C, m.Q, m.R, m.lo, m.hi);
    controller.set_epsilon(eps);
    controller.set_max_iterations(max_iter);
    for (unsigned long t = 0; t < H; ++t) {
        dlib::matrix<double, 2, 1> tg;
        tg = targets[2 * t], targets[2 * t + 1];
        controller.set_target(tg, t);
    }
    dlib::matrix<double, 2, 1> x;
    x = x0[0], x0[1];
    dlib::matrix<double, I, 1> a = controller(x);
    for (long j = 0; j < I; ++j) u0[j] = a(j);
}

template <long I, unsigned long H>
int general_all(long n, const double* A, const double* B, const double* C, const double* Q,
                const double* R, const double* lo, const double* hi, const double* x0,
                const double* targets, double eps, unsigned long max_iter, double* u0) {
    for (long k = 0; k < n; ++k) {
        Model<I> m = load_model<I>(A + 4 * k, B + 2 * I * k, C + 2 * k, Q + 2 * k, R + I * k,
                                   lo + I * k, hi + I * k);
        general_one<I, H>(m, x0 + 2 * k, targets + 2 * H * k, eps, max_iter, u0 + I * k);
    }
    return 0;
}

// ---- closed-loop rollout on ONE controller object (warm start, target shift) -------------
template <long I, unsigned long H>
int rollout(int steps, const double* A, const double* B, const double* C, const double* Q,
            const double* R, const double* lo, const double* hi, const double* x0,
            const double* targets0, const double* new_last_targets, double eps,
            unsigned long max_iter, double* controls_out, double* states_out) {
    Model<I> m = load_model<I>(A, B, C, Q, R, lo, hi);
    dlib::mpc<2, I, H> controller(m.A, m.B, m.C, m.Q, m.R, m.lo, m.hi);
    controller.set_epsilon(eps);
    controller.set_max_iterations(max_iter);
    for (unsigned long t = 0; t < H; ++t) {
        dlib::matrix<double, 2, 1> tg;
        tg = targets0[2 * t], targets0[2 * t + 1];
        controller.set_target(tg, t);
    }
    dlib::matrix<double, 2, 1> x;
    x = x0[0], x0[1];
    for (int s = 0; s < steps; ++s) {
        if (new_last_targets && s > 0) {
            // after operator() shifted the targets left (mpc.h:236-237) the caller supplies
            // the newly exposed last step (mpc_abstract.h: set_last_target)
            dlib::matrix<double, 2, 1> tg;
            tg = new_last_targets[2 * s], new_last_targets[2 * s + 1];
            controller.set_last_target(tg);
        }
        dlib::matrix<double, I, 1> u = controller(x);
        for (long j = 0; j < I; ++j) controls_out[I * s + j] = u(j);
        x = m.A * x + m.B * u + m.C;   // same plant update as dlib/test/mpc.cpp:314
        states_out[2 * s] = x(0);
        states_out[2 * s + 1] = x(1);
    }
    return 0;
}

}  // namespace

#define DLIBREF_FOR_H(X) X(4) X(5) X(10) X(20) X(30) X(40)

extern "C" {

// returns 0 ok, -1 unsupported H
int dlibref_solve_compact(int H, long n, int nthreads, const double* v, const double* dy,
                          const double* dphi, const double* weights4, double T, double l,
                          const double* lo2, const double* hi2, double eps,
                          unsigned long max_iter, double* out_front, double* out_rear) {
    switch (H) {
#define X(h)                                                                                 \
    case h:                                                                                  \
        return compact_threads<h>(n, nthreads, v, dy, dphi, weights4, T, l, lo2, hi2, eps,   \
                                  max_iter, out_front, out_rear);
        DLIBREF_FOR_H(X)
#undef X
    }
    return -1;
}

// AoS per-instance arrays: A[n][4] B[n][2*I] C[n][2] Q[n][2] R[n][I] lo[n][I] hi[n][I]
// x0[n][2] targets[n][H][2] -> u0[n][I]
int dlibref_solve_general(int I, int H, long n, const double* A, const double* B,
                          const double* C, const double* Q, const double* R, const double* lo,
                          const double* hi, const double* x0, const double* targets, double eps,
                          unsigned long max_iter, double* u0) {
#define X(h)                                                                                 \
    if (H == h && I == 2)                                                                    \
        return general_all<2, h>(n, A, B, C, Q, R, lo, hi, x0, targets, eps, max_iter, u0);  \
    if (H == h && I == 1)                                                                    \
        return general_all<1, h>(n, A, B, C, Q, R, lo, hi, x0, targets, eps, max_iter, u0);
    DLIBREF_FOR_H(X)
#undef X
    return -1;
}

int dlibref_rollout(int I, int H, int steps, const double* A, const double* B, const double* C,
                    const double* Q, const double* R, const double* lo, const double* hi,
                    const double* x0, const double* targets0, const double* new_last_targets,
                    double eps, unsigned long max_iter, double* controls_out,
                    double* states_out) {
#define X(h)                                                                                 \
    if (H == h && I == 2)                                                                    \
        return rollout<2, h>(steps, A, B, C, Q, R, lo, hi, x0, targets0, new_last_targets,   \
                             eps, max_iter, controls_out, states_out);                       \
    if (H == h && I == 1)                                                                    \
        return rollout<1, h>(steps, A, B, C, Q, R, lo, hi, x0, targets0, new_last_targets,   \
                             eps, max_iter, controls_out, states_out);
    DLIBREF_FOR_H(X)
#undef X
    return -1;
}

int dlibref_version(int* major, int* minor) {
    *major = DLIB_MAJOR_VERSION;
    *minor = DLIB_MINOR_VERSION;
    return 0;
}

}  // extern "C"
