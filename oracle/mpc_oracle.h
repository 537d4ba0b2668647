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

/* Arithmetic type.  The default build (liboracle_mpc.so) is the fp64 restatement that is pinned
 * bit for bit to real dlib.  -DMPC_ORACLE_REAL=float builds the SAME source with every value,
 * intermediate and constant typed float (liboracle_mpc_f32.so): the checker of the fp32 kernels.
 * dlib::mpc is fp64-only, so the float build has no reference counterpart to be pinned to
 * ("parity unpinned" for fp32): it states what "the same operation sequence in fp32" means, bit
 * for bit.  Scalar knobs (T, l, eps) stay double in the batch API and are rounded to the
 * arithmetic type once, on entry. */
#ifndef MPC_ORACLE_REAL
#define MPC_ORACLE_REAL double
#endif
typedef MPC_ORACLE_REAL mpc_real;

typedef struct mpc_oracle {
    int I;                 /* number of control inputs, 1 or 2 (S is fixed at 2) */
    int H;                 /* horizon, 1..MPC_ORACLE_MAX_H */
    mpc_real A[4];           /* 2x2 row-major */
    mpc_real B[4];           /* 2xI row-major */
    mpc_real C[2], Q[2];
    mpc_real R[2], lo[2], hi[2];
    mpc_real eps;            /* mpc.h:104  default 0.01 */
    unsigned long max_iter;/* mpc.h:103  default 10000 */
    unsigned long smo_iters;/* mpc.h:319 default 50 */
    mpc_real lambda;         /* mpc.h:116-123 */
    mpc_real Q_diag[MPC_ORACLE_MAX_H][2];
    mpc_real target[MPC_ORACLE_MAX_H][2];
    mpc_real controls[MPC_ORACLE_MAX_H][2];
    mpc_real v[MPC_ORACLE_MAX_H][2];      /* persists across calls like dlib's member (mpc.h:250) */
    unsigned long last_iters;           /* value of `iter` when solve_linear_mpc left its loop */
} mpc_oracle;

/* mpc.h:51-125.  B is 2xI row-major.  v[] is zeroed (dlib leaves it uninitialised). */
void mpc_oracle_init(mpc_oracle* s, int I, int H, const mpc_real* A, const mpc_real* B,
                     const mpc_real* C, const mpc_real* Q, const mpc_real* R, const mpc_real* lo,
                     const mpc_real* hi);
/* mpc.h:157-163 */
void mpc_oracle_set_target_all(mpc_oracle* s, const mpc_real* val2);
/* mpc.h:142-155 */
void mpc_oracle_set_target(mpc_oracle* s, const mpc_real* val2, int time);
/* mpc.h:216-240: warm-start shift, solve, target shift; writes controls[0] to u0[I] */
void mpc_oracle_step(mpc_oracle* s, const mpc_real* x0, mpc_real* u0);

/* Batch drivers.  Same argument meaning as the dlibref_* functions of ref_dlib_harness.cpp;
 * iters may be NULL; controls_out (n x H x I, optional) receives the full control sequence.
 * Return 0, or -1 for unsupported I/H. */
int mpc_oracle_solve_compact(int H, long n, int nthreads, const mpc_real* v, const mpc_real* dy,
                             const mpc_real* dphi, const mpc_real* weights4, double T, double l,
                             const mpc_real* lo2, const mpc_real* hi2, double eps,
                             unsigned long max_iter, unsigned long smo_iters,
                             mpc_real* out_front, mpc_real* out_rear, int* iters);

int mpc_oracle_solve_general(int I, int H, long n, int nthreads, const mpc_real* A,
                             const mpc_real* B, const mpc_real* C, const mpc_real* Q, const mpc_real* R,
                             const mpc_real* lo, const mpc_real* hi, const mpc_real* x0,
                             const mpc_real* targets, const mpc_real* controls_in, double eps,
                             unsigned long max_iter, unsigned long smo_iters, mpc_real* u0,
                             mpc_real* controls_out, int* iters);

/* ... with dlib's accelerated-gradient memory v (mpc.h:250; n x H x I) carried in and out */
int mpc_oracle_solve_general_state(int I, int H, long n, int nthreads, const mpc_real* A,
                                   const mpc_real* B, const mpc_real* C, const mpc_real* Q,
                                   const mpc_real* R, const mpc_real* lo, const mpc_real* hi,
                                   const mpc_real* x0, const mpc_real* targets,
                                   const mpc_real* controls_in, const mpc_real* v_in, double eps,
                                   unsigned long max_iter, unsigned long smo_iters, mpc_real* u0,
                                   mpc_real* controls_out, mpc_real* v_out, int* iters);

int mpc_oracle_rollout(int I, int H, int steps, const mpc_real* A, const mpc_real* B,
                       const mpc_real* C, const mpc_real* Q, const mpc_real* R, const mpc_real* lo,
                       const mpc_real* hi, const mpc_real* x0, const mpc_real* targets0,
                       const mpc_real* new_last_targets, double eps, unsigned long max_iter,
                       unsigned long smo_iters, mpc_real* controls_out, mpc_real* states_out,
                       int* iters);

#ifdef __cplusplus
}
#endif
#endif
