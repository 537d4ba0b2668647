// Host path of tpc_mpc_solve_one: one instance of the compact form solved on the calling CPU thread, in the LANE_FMA
// family's arithmetic (mpc_ub_host.h drives mpc_ub_model.h: the same IEEE operations the gfx950 kernels of that family
// execute -- tests/test_ub_gpu.py holds them to this code bit for bit).
//
// Why it exists.  The reference solves ONE horizon-4 problem per cycle() (include/trajectory_point_follower.h:48,
// src/trajectory_point_follower.cpp:366-380): ~5 us of dlib on a core.  A GPU wavefront cannot beat a serial chain
// that short -- the resident wavefront needs ~10 us at N = 4 (2.8 us of transport plus ~80 iterations at a lone
// wavefront's issue rate) -- and SURVEY.md section 8(b) asks for a solve_one "usable from any single thread without a
// GPU".  So: a handle created with TPC_MPC_DEVICE_NONE has this path and nothing else, and a GPU handle takes it for
// horizons up to TPC_MPC_OPT_HOST_SOLVE_ONE.  It is the product's own arithmetic, not the checker's: nothing here (or
// anywhere in the product) touches oracle/ or tests/.
//
// This translation unit is built with -mfma (fma() must be the instruction); tpc_mpc_api.cpp asks the CPU before it
// calls in here (tpc::host_path_usable).
#include "../../include/tpc_mpc.h"
#include "mpc_ub_host.h"

namespace tpc {

namespace {

template <int H, bool EQB>
int solve(const tpc_mpc_params* p, double v, double dy, double dphi, double* front, double* rear, int* iters, unsigned* flags) {
    const double q[2] = {p->weight_y, p->weight_phi}, r[2] = {p->weight_steering_front, p->weight_steering_rear};
    // (-1: the screen of the select-free stop test is applied to this one instance; ub_cd_kernel decides the same per batch)
    ub::host_solve_compact<double, H, EQB>(v, dy, dphi, q, r, p->step_size, p->wheelbase, p->lower, p->upper, p->eps,
                                           (unsigned long)p->max_iter, (unsigned long)p->smo_iters, -1, front, rear, iters, flags);
    return 0;
}

}  // namespace

// 0 = solved; -1 = not a request this path takes (horizon without a specialised driver).  The caller has checked
// dtype (fp64), the bounds (finite, upper > lower: the unit box) and the CPU (fma).
int host_solve_one(const tpc_mpc_params* p, double v, double dy, double dphi, double* front, double* rear, int* iters,
                   unsigned* flags) {
    const bool eqb = p->lower[0] == p->lower[1] && p->upper[0] == p->upper[1];
    *flags = 0;
    switch (p->horizon) {
#define X(h) case h: return eqb ? solve<h, true>(p, v, dy, dphi, front, rear, iters, flags) : solve<h, false>(p, v, dy, dphi, front, rear, iters, flags);
        X(4) X(5) X(10) X(20) X(30) X(40)
#undef X
    }
    return -1;
}

}  // namespace tpc
