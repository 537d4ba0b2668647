// Element-wise helpers of tpc_mpc_rollout: what the caller of dlib::mpc does between two
// operator() calls in a closed loop (reference: dlib_files/dlib/test/mpc.cpp:301-316) plus the
// target shift operator() performs itself (mpc.h:236-237).  One thread per instance, SoA,
// coalesced; HBM-bound and tiny next to the solves.
#include "mpc_internal.h"

namespace tpc {

template <typename T>
__global__ void rollout_step_kernel(RolloutStepArgs a) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= a.n) return;
    const int64_t ld = a.ld;
    const T* A = (const T*)a.A + k;
    const T* B = (const T*)a.B + k;
    const T* Cc = (const T*)a.C + k;
    T* x = (T*)a.x + k;
    const T* u = (const T*)a.controls + k;   // controls[0](j) at component j
    // record u0 and advance the plant: x <- A*x + B*u + C  (test/mpc.cpp:314)
    T bu0 = B[0] * u[0], bu1 = B[(int64_t)a.I * ld] * u[0];
    if (a.I == 2) { bu0 = bu0 + B[ld] * u[ld]; bu1 = bu1 + B[3 * ld] * u[ld]; }
    const T x0 = x[0], x1 = x[ld];
    const T n0 = ((A[0] * x0 + A[ld] * x1) + bu0) + Cc[0];
    const T n1 = ((A[2 * ld] * x0 + A[3 * ld] * x1) + bu1) + Cc[ld];
    x[0] = n0; x[ld] = n1;
    const int64_t lo = a.ld_out;
    for (int j = 0; j < a.I; ++j) ((T*)a.controls_out)[((int64_t)a.step * a.I + j) * lo + k] = u[(int64_t)j * ld];
    if (a.states_out) {
        ((T*)a.states_out)[((int64_t)a.step * 2) * lo + k] = n0;
        ((T*)a.states_out)[((int64_t)a.step * 2 + 1) * lo + k] = n1;
    }
    if (a.iters_out && a.iters_step) a.iters_out[(int64_t)a.step * lo + k] = a.iters_step[k];
    // operator()'s target shift (mpc.h:236-237), then the caller's set_last_target for the next call
    T* t = (T*)a.targets + k;
    for (int i = 1; i < a.H; ++i) {
        t[(int64_t)(2 * (i - 1)) * ld] = t[(int64_t)(2 * i) * ld];
        t[(int64_t)(2 * (i - 1) + 1) * ld] = t[(int64_t)(2 * i + 1) * ld];
    }
    if (a.new_last_targets && a.step + 1 < a.steps) {
        const T* nl = (const T*)a.new_last_targets + k;
        t[(int64_t)(2 * (a.H - 1)) * ld] = nl[(2 * ((int64_t)a.step + 1)) * a.ld_nlt];
        t[(int64_t)(2 * (a.H - 1) + 1) * ld] = nl[(2 * ((int64_t)a.step + 1) + 1) * a.ld_nlt];
    }
}

hipError_t launch_rollout_step(int dtype, const RolloutStepArgs& a, hipStream_t s) {
    const int block = 256;
    const unsigned grid = (unsigned)((a.n + block - 1) / block);
    if (dtype == 0) hipLaunchKernelGGL(rollout_step_kernel<double>, dim3(grid), dim3(block), 0, s, a);
    else hipLaunchKernelGGL(rollout_step_kernel<float>, dim3(grid), dim3(block), 0, s, a);
    return hipGetLastError();
}

}  // namespace tpc
