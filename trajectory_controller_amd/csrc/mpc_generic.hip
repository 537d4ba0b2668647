// Any-horizon fallback: dlib::mpc<2,I,H> is a template, so a maintainer can set MPC_HORIZON
// (reference: include/trajectory_point_follower.h:48) to any value; the specialised LANE / WAVE
// kernels exist for the horizons of tpc_mpc_supported_horizons() only.  This kernel takes every other
// horizon 1 <= H <= 64 with H as a RUN-TIME value: one lane per instance like the LANE family, the
// same arithmetic in the same order (mpc_model.h policies; fp64 results equal dlib's bit for bit),
// but the per-instance arrays (controls, forward pass / gradient, linear term, momentum, Q_diag) live
// in a global-memory workspace laid out [slot][instance] -- every access of a wavefront is 64
// consecutive elements -- instead of registers, loops are not unrolled, and there is no work queue:
// a wavefront runs until its slowest lane is done.  Several times slower than a specialised kernel;
// it is the difference between "slower" and TPC_MPC_ERR_BAD_HORIZON.
#include "mpc_model.h"

namespace tpc {

namespace {

template <typename T> struct GenericWs {
    T *u, *w, *mm, *v, *qd;   // each [2*H][n], slot q of instance k at base[q*n + k]
};

template <typename T, int I, class Model, class Args>
__global__ __launch_bounds__(64) void lane_generic_kernel(Args g, Knobs kn, int H, GenericWs<T> ws) {
    const int64_t n = g.n;
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    T* U = ws.u + k;
    T* W = ws.w + k;
    T* MM = ws.mm + k;
    T* V = ws.v + k;
    T* QD = ws.qd + k;
    auto at = [n](T* base, int q) -> T& { return base[(int64_t)q * n]; };

    Model m;
    m.load(g, k);
    const bool nonfinite = m.nonfinite();
    const bool badmodel = m.invalid();

    // ---- controller state in (mpc.h:105-112 zeros, or the caller's with the warm-start shift, :231-232)
    for (int i = 0; i < H; ++i)
        for (int j = 0; j < 2; ++j) { at(U, 2 * i + j) = (T)0; at(V, 2 * i + j) = (T)0; }
    if constexpr (std::is_same<Args, GeneralArgs>::value) {
        if (g.controls) {
            const T* cp = (const T*)g.controls + k;
            for (int i = 0; i < H; ++i)
                for (int j = 0; j < I; ++j) {
                    const int src = (g.shift_controls && i + 1 < H) ? i + 1 : i;
                    at(U, 2 * i + j) = cp[(int64_t)(src * I + j) * g.ld];
                }
        }
        if (g.v) {
            const T* vp = (const T*)g.v + k;
            for (int i = 0; i < H; ++i)
                for (int j = 0; j < I; ++j) at(V, 2 * i + j) = vp[(int64_t)(i * I + j) * g.ld];
        }
    }

    // ---- constructor quantities (mpc.h:116-123), as ctor_lambda_qdiag with a run-time horizon
    T sumR = m.R(0);
    if (I == 2) sumR = sumR + m.R(I - 1);
    T lambda = sumR * (T)H;
    {
        T t00 = m.Q(0), t01 = (T)0, t10 = (T)0, t11 = m.Q(1);
        for (int cidx = 0; cidx < H; ++cidx) {
            T tr = (T)0;
#pragma unroll
            for (int r_ = 0; r_ < I; ++r_) {
                const T w0 = m.B(0, r_) * t00 + m.B(1, r_) * t10;
                const T w1 = m.B(0, r_) * t01 + m.B(1, r_) * t11;
                const T p = w0 * m.B(0, r_) + w1 * m.B(1, r_);
                at(QD, 2 * (H - cidx - 1) + r_) = p;
                tr = (r_ == 0) ? p : tr + p;
            }
            lambda = lambda + tr;
            const T u00 = m.A(0, 0) * t00 + m.A(1, 0) * t10, u01 = m.A(0, 0) * t01 + m.A(1, 0) * t11;
            const T u10 = m.A(0, 1) * t00 + m.A(1, 1) * t10, u11 = m.A(0, 1) * t01 + m.A(1, 1) * t11;
            const T n00 = (u00 * m.A(0, 0) + u01 * m.A(1, 0)) + m.Q(0);
            const T n01 = (u00 * m.A(0, 1) + u01 * m.A(1, 1)) + (T)0;
            const T n10 = (u10 * m.A(0, 0) + u11 * m.A(1, 0)) + (T)0;
            const T n11 = (u10 * m.A(0, 1) + u11 * m.A(1, 1)) + m.Q(1);
            t00 = n00; t01 = n01; t10 = n10; t11 = n11;
        }
    }
    // ---- linear term MM (mpc.h:258-266), as linear_term_fn
    {
        T m0 = (m.A(0, 0) * m.x0(0) + m.A(0, 1) * m.x0(1)) + m.C(0);
        T m1 = (m.A(1, 0) * m.x0(0) + m.A(1, 1) * m.x0(1)) + m.C(1);
        for (int i = 0; i < H; ++i) {
            if (i > 0) {
                const T n0 = (m.A(0, 0) * m0 + m.A(0, 1) * m1) + m.C(0);
                const T n1 = (m.A(1, 0) * m0 + m.A(1, 1) * m1) + m.C(1);
                m0 = n0; m1 = n1;
            }
            at(W, 2 * i) = (m0 - m.target(i, 0)) * m.Q(0);
            at(W, 2 * i + 1) = (m1 - m.target(i, 1)) * m.Q(1);
        }
        T n0 = at(W, 2 * (H - 1)), n1 = at(W, 2 * (H - 1) + 1);
        for (int i = H - 1; i >= 0; --i) {
            if (i < H - 1) {
                const T t0 = at(W, 2 * i) + (m.A(0, 0) * n0 + m.A(1, 0) * n1);
                const T t1 = at(W, 2 * i + 1) + (m.A(0, 1) * n0 + m.A(1, 1) * n1);
                n0 = t0; n1 = t1;
            }
            for (int j = 0; j < I; ++j) at(MM, 2 * i + j) = m.B(0, j) * n0 + m.B(1, j) * n1;
        }
    }
    const T eps = (T)kn.eps;
    const T inv_lambda = (T)1.0 / lambda;                 // mpc.h:342
    const T sq = tsqrt(lambda);
    const T beta = (sq - (T)1) / (sq + (T)1);             // mpc.h:343

    uint32_t iter = 0;
    bool capped = true;
    if ((Model::kScreen && nonfinite) || badmodel) {
        capped = false;
    } else {
        for (; iter < kn.max_iter; ++iter) {
            // gradient by dlib's recurrences (mpc.h:275-283), as gradient_fn
            T m0, m1;
            {
                const T ui[2] = {at(U, 0), at(U, 1)};
                m.first(m0, m1, ui);
            }
            at(W, 0) = m0; at(W, 1) = m1;
            for (int i = 1; i < H; ++i) {
                const T ui[2] = {at(U, 2 * i), at(U, 2 * i + 1)};
                m.fwd(m0, m1, ui);
                at(W, 2 * i) = m0; at(W, 2 * i + 1) = m1;
            }
            T n0 = at(W, 2 * (H - 1)) * m.Q(0), n1 = at(W, 2 * (H - 1) + 1) * m.Q(1);
            // the backward pass yields df[i], and the arg-max over free variables is taken on the way.
            // dlib scans i ascending with a strict '>' (mpc.h:289-309): the LOWEST index wins a tie, so
            // scanning downwards the later (lower) index must win ties: '>='.  NaN loses either way.
            T max_df = (T)0;
            int best = 0;
            for (int i = H - 1; i >= 0; --i) {
                if (i < H - 1) m.bwd(n0, n1, at(W, 2 * i), at(W, 2 * i + 1));
                for (int j = I - 1; j >= 0; --j) {
                    const T uu = at(U, 2 * i + j);
                    const T dd = (at(MM, 2 * i + j) + m.btm(j, n0, n1)) + uu * m.R(j);
                    at(W, 2 * i + j) = dd;
                    const bool blocked = (uu <= m.lo(j) && dd > (T)0) || (uu >= m.hi(j) && dd < (T)0);
                    const T mag = tabs(dd);
                    if (!blocked && mag >= max_df && mag > (T)0) { max_df = mag; best = 2 * i + j; }
                }
                if (I == 1) at(W, 2 * i + 1) = (T)0;
            }
            if (max_df < eps) { capped = false; break; }                          // mpc.h:310-311
            if (iter < kn.smo_iters) {                                              // mpc.h:319-335
                const T qdv = at(QD, best);
                if (qdv != (T)0) {                                                  // mpc.h:322
                    const int bj = best & 1;
                    T nu = -(at(W, best) - qdv * at(U, best)) / qdv;                // mpc.h:325
                    at(U, best) = put_in_range(m.lo(bj), m.hi(bj), nu);             // mpc.h:326
                    if (iter + 1 == kn.smo_iters)                                   // mpc.h:330-334
                        for (int i = 0; i < H; ++i)
                            for (int j = 0; j < I; ++j) at(V, 2 * i + j) = at(U, 2 * i + j);
                }
            } else {                                                                // mpc.h:336-345
                for (int i = 0; i < H; ++i)
                    for (int j = 0; j < I; ++j) {
                        const int q = 2 * i + j;
                        const T v_old = at(V, q);
                        const T vn = clamp3(at(U, q) - inv_lambda * at(W, q), m.lo(j), m.hi(j));
                        at(V, q) = vn;
                        at(U, q) = clamp3(vn + beta * (vn - v_old), m.lo(j), m.hi(j));
                    }
            }
        }
    }

    // ---- outputs
    if constexpr (std::is_same<Args, CompactArgs>::value) {
        ((T*)g.front)[k] = at(U, 0);
        ((T*)g.rear)[k] = at(U, 1);
    } else {
        for (int j = 0; j < I; ++j) ((T*)g.u0)[(int64_t)j * g.ld + k] = at(U, j);
        if (g.controls) {
            T* cp = (T*)g.controls + k;
            for (int i = 0; i < H; ++i)
                for (int j = 0; j < I; ++j) cp[(int64_t)(i * I + j) * g.ld] = at(U, 2 * i + j);
        }
        if (g.v) {
            T* vp = (T*)g.v + k;
            for (int i = 0; i < H; ++i)
                for (int j = 0; j < I; ++j) vp[(int64_t)(i * I + j) * g.ld] = at(V, 2 * i + j);
        }
    }
    if (g.iters) g.iters[k] = (int32_t)iter;
    uint32_t f = 0;
    if (nonfinite) f |= 0x1u;
    if (badmodel) f |= 0x4u;
    if (capped) f |= 0x2u;
    raise_flags(g.flags, f);
}

template <typename T, int I, class Model, class Args>
hipError_t run_generic(const Args& a, const Knobs& k, int H, void* scratch, hipStream_t s) {
    if (a.n <= 0) return hipSuccess;
    const int64_t slab = (int64_t)2 * H * a.n;
    GenericWs<T> ws;
    ws.u = (T*)scratch;
    ws.w = ws.u + slab;
    ws.mm = ws.w + slab;
    ws.v = ws.mm + slab;
    ws.qd = ws.v + slab;
    const unsigned grid = (unsigned)((a.n + kWave - 1) / kWave);
    hipLaunchKernelGGL((lane_generic_kernel<T, I, Model, Args>), dim3(grid), dim3(kWave), 0, s, a, k, H, ws);
    return hipGetLastError();
}

}  // namespace

int64_t generic_scratch_bytes(int H, int dtype, int64_t n) { return (int64_t)5 * 2 * H * n * (dtype == 0 ? 8 : 4); }

hipError_t generic_compact(int dtype, int H, const CompactArgs& a, const Knobs& k, void* scratch, hipStream_t s) {
    if (dtype == 0) return run_generic<double, 2, CompactModel<double>, CompactArgs>(a, k, H, scratch, s);
    return run_generic<float, 2, CompactModel<float>, CompactArgs>(a, k, H, scratch, s);
}

hipError_t generic_general(int dtype, int I, int H, const GeneralArgs& a, const Knobs& k, void* scratch, hipStream_t s) {
    if (dtype == 0) {
        if (I == 2) return run_generic<double, 2, GeneralModel<double, 2>, GeneralArgs>(a, k, H, scratch, s);
        return run_generic<double, 1, GeneralModel<double, 1>, GeneralArgs>(a, k, H, scratch, s);
    }
    if (I == 2) return run_generic<float, 2, GeneralModel<float, 2>, GeneralArgs>(a, k, H, scratch, s);
    return run_generic<float, 1, GeneralModel<float, 1>, GeneralArgs>(a, k, H, scratch, s);
}

}  // namespace tpc
