/* oracle/mpc_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C, IEEE fp64, no FMA contraction) of the reference's MPC QP path:
 * dlib::mpc<S=2, I in {1,2}, H> (reference: dlib_files/dlib/control/mpc.h:51-125 ctor,
 * :142-163 targets, :216-240 operator(), :253-347 solve_linear_mpc) as it is driven by
 * TrajectoryPointController::mpcControllerTobi (reference:
 * src/trajectory_point_follower.cpp:301-389).
 *
 * Parity status: PINNED.  The restatement is checked bit-for-bit against the real dlib 18.18
 * headers compiled in the build container (oracle/_ref/libdlib_mpc_ref.so, see
 * oracle/Makefile) and against the golden vectors committed under tests/golden/, which were
 * generated from that same real-dlib build (tests/golden/make_golden.py).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this code.
 * The product library (libtpc_mpc.so) never links, loads or calls it.
 */
#ifndef TPC_MPC_ORACLE_H
#define TPC_MPC_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

#define MPC_ORACLE_MAX_H 64

typedef struct mpc_oracle {
    int I;                 /* number of control inputs, 1 or 2 (S is fixed at 2) */
    int H;                 /* horizon, 1..MPC_ORACLE_MAX_H */
    double A[4];           /* 2x2 row-major */
    double B[4];           /* 2xI row-major */
    double C[2], Q[2];
    double R[2], lo[2], hi[2];
    double eps;            /* mpc.h:104  default 0.01 */
    unsigned long max_iter;/* mpc.h:103  default 10000 */
    unsigned long smo_iters;/* mpc.h:319 default 50 */
    double lambda;         /* mpc.h:116-123 */
    double Q_diag[MPC_ORACLE_MAX_H][2];
    double target[MPC_ORACLE_MAX_H][2];
    double controls[MPC_ORACLE_MAX_H][2];
    double v[MPC_ORACLE_MAX_H][2];      /* persists across calls like dlib's member (mpc.h:250) */
    unsigned long last_iters;           /* value of `iter` when solve_linear_mpc left its loop */
} mpc_oracle;

/* mpc.h:51-125.  B is 2xI row-major.  v[] is zeroed (dlib leaves it uninitialised). */
void mpc_oracle_init(mpc_oracle* s, int I, int H, const double* A, const double* B,
                     const double* C, const double* Q, const double* R, const double* lo,
                     const double* hi);
/* mpc.h:157-163 */
void mpc_oracle_set_target_all(mpc_oracle* s, const double* val2);
/* mpc.h:142-155 */
void mpc_oracle_set_target(mpc_oracle* s, const double* val2, int time);
/* mpc.h:216-240: warm-start shift, solve, target shift; writes controls[0] to u0[I] */
void mpc_oracle_step(mpc_oracle* s, const double* x0, double* u0);

/* Batch drivers.  Same argument meaning as the dlibref_* functions of ref_dlib_harness.cpp;
 * iters may be NULL; controls_out (n x H x I, optional) receives the full control sequence.
 * Return 0, or -1 for unsupported I/H. */
int mpc_oracle_solve_compact(int H, long n, int nthreads, const double* v, const double* dy,
                             const double* dphi, const double* weights4, double T, double l,
                             const double* lo2, const double* hi2, double eps,
                             unsigned long max_iter, unsigned long smo_iters,
                             double* out_front, double* out_rear, int* iters);

int mpc_oracle_solve_general(int I, int H, long n, int nthreads, const double* A,
                             const double* B, const double* C, const double* Q, const double* R,
                             const double* lo, const double* hi, const double* x0,
                             const double* targets, const double* controls_in, double eps,
                             unsigned long max_iter, unsigned long smo_iters, double* u0,
                             double* controls_out, int* iters);

int mpc_oracle_rollout(int I, int H, int steps, const double* A, const double* B,
                       const double* C, const double* Q, const double* R, const double* lo,
                       const double* hi, const double* x0, const double* targets0,
                       const double* new_last_targets, double eps, unsigned long max_iter,
                       unsigned long smo_iters, double* controls_out, double* states_out,
                       int* iters);

#ifdef __cplusplus
}
#endif
#endif
