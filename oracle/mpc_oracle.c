/* oracle/mpc_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.  See mpc_oracle.h.
 *
 * Plain-C restatement of dlib::mpc (dlib 18.18, reference: dlib_files/dlib/control/mpc.h).
 * Every function cites the reference lines it follows.  Arithmetic association follows
 * dlib's expression templates: a matrix product element is lhs(r,0)*rhs(0,c) then
 * += lhs(r,k)*rhs(k,c) for k ascending (reference: dlib_files/dlib/matrix/matrix.h:43-61);
 * with S = 2 and I <= 2 every inner sum has at most two terms.  Build with
 * -ffp-contract=off: the reference build (no -march) emits no FMA.
 */
#include "mpc_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* sqrt / fabs of the arithmetic type (correctly rounded in either) */
#define MPC_IS_F64 (sizeof(mpc_real) == sizeof(double))
static inline mpc_real real_sqrt(mpc_real x) { return MPC_IS_F64 ? (mpc_real)sqrt((double)x) : (mpc_real)sqrtf((float)x); }
static inline mpc_real real_fabs(mpc_real x) { return MPC_IS_F64 ? (mpc_real)fabs((double)x) : (mpc_real)fabsf((float)x); }

/* reference: dlib_files/dlib/algs.h:716-752 */
static inline mpc_real put_in_range(mpc_real a, mpc_real b, mpc_real val) {
    if (a < b) {
        if (val < a) return a;
        else if (val > b) return b;
    } else {
        if (val < b) return b;
        else if (val > a) return a;
    }
    return val;
}

/* reference: dlib_files/dlib/matrix/matrix_utilities.h:2835-2846 (3-matrix clamp) */
static inline mpc_real clamp3(mpc_real val, mpc_real lower, mpc_real upper) {
    if (val <= upper) {
        if (lower <= val) return val;
        else return lower;
    }
    return upper;
}

/* reference: mpc.h:51-125 */
void mpc_oracle_init(mpc_oracle* s, int I, int H, const mpc_real* A, const mpc_real* B,
                     const mpc_real* C, const mpc_real* Q, const mpc_real* R, const mpc_real* lo,
                     const mpc_real* hi) {
    memset(s, 0, sizeof(*s));
    s->I = I;
    s->H = H;
    memcpy(s->A, A, 4 * sizeof(mpc_real));
    memcpy(s->B, B, 2 * I * sizeof(mpc_real));
    memcpy(s->C, C, 2 * sizeof(mpc_real));
    memcpy(s->Q, Q, 2 * sizeof(mpc_real));
    memcpy(s->R, R, I * sizeof(mpc_real));
    memcpy(s->lo, lo, I * sizeof(mpc_real));
    memcpy(s->hi, hi, I * sizeof(mpc_real));
    s->max_iter = 10000;   /* mpc.h:103 */
    s->eps = (mpc_real)0.01; /* mpc.h:104 */
    s->smo_iters = 50;     /* mpc.h:319 */
    /* target[i] = 0, controls[i] = 0 (mpc.h:105-112): done by memset */

    /* mpc.h:116  lambda = sum(R)*horizon */
    mpc_real sumR = 0;
    for (int j = 0; j < I; ++j) sumR += R[j];
    mpc_real lambda = sumR * (mpc_real)(unsigned long)H;
    /* mpc.h:117  temp = diagm(Q) */
    mpc_real T[2][2] = {{Q[0], (mpc_real)0}, {(mpc_real)0, Q[1]}};
    const mpc_real(*Am)[2] = (const mpc_real(*)[2])s->A;
    for (int c = 0; c < H; ++c) {
        /* trans(B)*temp*B, left-associated: W = trans(B)*temp (IxS), P = W*B (IxI) */
        mpc_real W[2][2], P[2][2];
        for (int r = 0; r < I; ++r)
            for (int k = 0; k < 2; ++k)
                W[r][k] = s->B[0 * I + r] * T[0][k] + s->B[1 * I + r] * T[1][k];
        for (int r = 0; r < I; ++r)
            for (int k = 0; k < I; ++k)
                P[r][k] = W[r][0] * s->B[0 * I + k] + W[r][1] * s->B[1 * I + k];
        /* mpc.h:120  lambda += trace(...) */
        mpc_real tr = 0;
        for (int r = 0; r < I; ++r) tr += P[r][r];
        lambda += tr;
        /* mpc.h:121  Q_diag[horizon-c-1] = diag(...) */
        for (int r = 0; r < I; ++r) s->Q_diag[H - c - 1][r] = P[r][r];
        /* mpc.h:122  temp = trans(A)*temp*A + diagm(Q) */
        mpc_real U[2][2], Tn[2][2];
        for (int r = 0; r < 2; ++r)
            for (int k = 0; k < 2; ++k) U[r][k] = Am[0][r] * T[0][k] + Am[1][r] * T[1][k];
        for (int r = 0; r < 2; ++r)
            for (int k = 0; k < 2; ++k)
                Tn[r][k] = (U[r][0] * Am[0][k] + U[r][1] * Am[1][k]) + (r == k ? Q[r] : (mpc_real)0);
        memcpy(T, Tn, sizeof(T));
    }
    s->lambda = lambda;
}

/* reference: mpc.h:157-163 */
void mpc_oracle_set_target_all(mpc_oracle* s, const mpc_real* val2) {
    for (int i = 0; i < s->H; ++i) {
        s->target[i][0] = val2[0];
        s->target[i][1] = val2[1];
    }
}

/* reference: mpc.h:142-155 */
void mpc_oracle_set_target(mpc_oracle* s, const mpc_real* val2, int time) {
    s->target[time][0] = val2[0];
    s->target[time][1] = val2[1];
}

/* reference: mpc.h:253-347.  I is passed as a literal so the compiler specialises the loops. */
static inline __attribute__((always_inline)) void solve_linear_mpc(mpc_oracle* s,
                                                                   const mpc_real* x0,
                                                                   const int I) {
    const int H = s->H;
    const mpc_real(*A)[2] = (const mpc_real(*)[2])s->A;
    const mpc_real* B = s->B; /* B[r*I + j] */
    mpc_real M[MPC_ORACLE_MAX_H][2], MM[MPC_ORACLE_MAX_H][2], df[MPC_ORACLE_MAX_H][2];
    mpc_real(*u)[2] = s->controls;
    mpc_real(*v)[2] = s->v;

    /* mpc.h:258-260  M[0] = A*x0 + C ; M[i] = A*M[i-1] + C */
    for (int r = 0; r < 2; ++r) M[0][r] = (A[r][0] * x0[0] + A[r][1] * x0[1]) + s->C[r];
    for (int i = 1; i < H; ++i)
        for (int r = 0; r < 2; ++r)
            M[i][r] = (A[r][0] * M[i - 1][0] + A[r][1] * M[i - 1][1]) + s->C[r];
    /* mpc.h:261-262  M[i] = diagm(Q)*(M[i]-target[i]) */
    for (int i = 0; i < H; ++i)
        for (int r = 0; r < 2; ++r) M[i][r] = (M[i][r] - s->target[i][r]) * s->Q[r];
    /* mpc.h:263-264  M[i] += trans(A)*M[i+1] */
    for (int i = H - 2; i >= 0; --i)
        for (int r = 0; r < 2; ++r)
            M[i][r] = M[i][r] + (A[0][r] * M[i + 1][0] + A[1][r] * M[i + 1][1]);
    /* mpc.h:265-266  MM[i] = trans(B)*M[i] */
    for (int i = 0; i < H; ++i)
        for (int j = 0; j < I; ++j) MM[i][j] = B[0 * I + j] * M[i][0] + B[1 * I + j] * M[i][1];

    /* loop-invariant scalars of mpc.h:342-343 (pure functions of lambda) */
    const mpc_real inv_lambda = (mpc_real)1 / s->lambda;
    const mpc_real sq = real_sqrt(s->lambda);
    const mpc_real beta = (sq - 1) / (sq + 1);

    unsigned long iter = 0;
    for (; iter < s->max_iter; ++iter) {
        /* mpc.h:275-277  M[0] = B*u[0] ; M[i] = A*M[i-1] + B*u[i] */
        for (int r = 0; r < 2; ++r) {
            mpc_real bu = B[r * I + 0] * u[0][0];
            if (I == 2) bu = bu + B[r * I + 1] * u[0][1];
            M[0][r] = bu;
        }
        for (int i = 1; i < H; ++i)
            for (int r = 0; r < 2; ++r) {
                mpc_real bu = B[r * I + 0] * u[i][0];
                if (I == 2) bu = bu + B[r * I + 1] * u[i][1];
                M[i][r] = (A[r][0] * M[i - 1][0] + A[r][1] * M[i - 1][1]) + bu;
            }
        /* mpc.h:278-279 */
        for (int i = 0; i < H; ++i)
            for (int r = 0; r < 2; ++r) M[i][r] = M[i][r] * s->Q[r];
        /* mpc.h:280-281 */
        for (int i = H - 2; i >= 0; --i)
            for (int r = 0; r < 2; ++r)
                M[i][r] = M[i][r] + (A[0][r] * M[i + 1][0] + A[1][r] * M[i + 1][1]);
        /* mpc.h:282-283  df[i] = MM[i] + trans(B)*M[i] + diagm(R)*u[i] */
        for (int i = 0; i < H; ++i)
            for (int j = 0; j < I; ++j)
                df[i][j] = (MM[i][j] + (B[0 * I + j] * M[i][0] + B[1 * I + j] * M[i][1])) +
                           u[i][j] * s->R[j];

        /* mpc.h:289-311 */
        mpc_real max_df = 0;
        int max_t = 0, max_v = 0;
        for (int i = 0; i < H; ++i)
            for (int j = 0; j < I; ++j)
                if (!((u[i][j] <= s->lo[j] && df[i][j] > 0) ||
                      (u[i][j] >= s->hi[j] && df[i][j] < 0))) {
                    if (real_fabs(df[i][j]) > max_df) {
                        max_df = real_fabs(df[i][j]);
                        max_t = i;
                        max_v = j;
                    }
                }
        if (max_df < s->eps) break;

        if (iter < s->smo_iters) { /* mpc.h:320-335 */
            const mpc_real qd = s->Q_diag[max_t][max_v];
            if (qd == 0) continue;
            u[max_t][max_v] = -(df[max_t][max_v] - qd * u[max_t][max_v]) / qd;
            u[max_t][max_v] = put_in_range(s->lo[max_v], s->hi[max_v], u[max_t][max_v]);
            if (iter + 1 == s->smo_iters)
                for (int i = 0; i < H; ++i)
                    for (int j = 0; j < I; ++j) v[i][j] = u[i][j];
        } else { /* mpc.h:336-345 */
            for (int i = 0; i < H; ++i)
                for (int j = 0; j < I; ++j) {
                    const mpc_real v_old = v[i][j];
                    v[i][j] = clamp3(u[i][j] - inv_lambda * df[i][j], s->lo[j], s->hi[j]);
                    u[i][j] = clamp3(v[i][j] + beta * (v[i][j] - v_old), s->lo[j], s->hi[j]);
                }
        }
    }
    s->last_iters = iter;
}

/* reference: mpc.h:216-240 */
void mpc_oracle_step(mpc_oracle* s, const mpc_real* x0, mpc_real* u0) {
    for (int i = 1; i < s->H; ++i) { /* mpc.h:231-232 */
        s->controls[i - 1][0] = s->controls[i][0];
        s->controls[i - 1][1] = s->controls[i][1];
    }
    if (s->I == 2) solve_linear_mpc(s, x0, 2);
    else solve_linear_mpc(s, x0, 1);
    for (int i = 1; i < s->H; ++i) { /* mpc.h:236-237 */
        s->target[i - 1][0] = s->target[i][0];
        s->target[i - 1][1] = s->target[i][1];
    }
    for (int j = 0; j < s->I; ++j) u0[j] = s->controls[0][j];
}

/* ---------------------------------------------------------------------------------------- */
/* batch drivers                                                                            */

typedef struct {
    int kind; /* 0 compact, 1 general */
    int I, H, tid, nthreads;
    long n;
    /* compact */
    const mpc_real *v, *dy, *dphi, *w;
    mpc_real T, l;
    mpc_real *out_front, *out_rear;
    /* general */
    const mpc_real *A, *B, *C, *Q, *R, *x0, *targets, *controls_in, *v_in;
    mpc_real *u0, *controls_out, *v_out;
    /* common */
    const mpc_real *lo, *hi;
    mpc_real eps;
    unsigned long max_iter, smo_iters;
    int* iters;
} job_t;

/* reference: src/trajectory_point_follower.cpp:326-384 (mpcControllerTobi) */
static void compact_one(const job_t* jb, long k, mpc_oracle* s) {
    const mpc_real T = jb->T, l = jb->l, vk = jb->v[k];
    const mpc_real A[4] = {1, T * vk, 0, 1};                       /* :326-327 */
    const mpc_real B[4] = {0, T * vk, T * vk / l, -T * vk / l};    /* :329-330 */
    const mpc_real C[2] = {0, 0};                                  /* :332-333 */
    const mpc_real Q[2] = {jb->w[0], jb->w[1]};                    /* :359-360 */
    const mpc_real R[2] = {jb->w[2], jb->w[3]};                    /* :362-363 */
    mpc_oracle_init(s, 2, jb->H, A, B, C, Q, R, jb->lo, jb->hi); /* :366 */
    const mpc_real target[2] = {jb->dy[k], jb->dphi[k]};           /* :368-371 */
    mpc_oracle_set_target_all(s, target);
    s->eps = jb->eps;
    s->max_iter = jb->max_iter;
    s->smo_iters = jb->smo_iters;
    const mpc_real x0[2] = {0, 0};                                 /* :377-378 */
    mpc_real u0[2];
    mpc_oracle_step(s, x0, u0);                                  /* :380 */
    jb->out_front[k] = u0[0];                                    /* :383-384 */
    jb->out_rear[k] = u0[1];
    if (jb->iters) jb->iters[k] = (int)s->last_iters;
}

static void general_one(const job_t* jb, long k, mpc_oracle* s) {
    const int I = jb->I, H = jb->H;
    mpc_oracle_init(s, I, H, jb->A + 4 * k, jb->B + 2 * I * k, jb->C + 2 * k, jb->Q + 2 * k,
                    jb->R + I * k, jb->lo + I * k, jb->hi + I * k);
    s->eps = jb->eps;
    s->max_iter = jb->max_iter;
    s->smo_iters = jb->smo_iters;
    for (int t = 0; t < H; ++t) mpc_oracle_set_target(s, jb->targets + (2 * H * k + 2 * t), t);
    if (jb->controls_in)
        for (int t = 0; t < H; ++t)
            for (int j = 0; j < I; ++j)
                s->controls[t][j] = jb->controls_in[(long)H * I * k + t * I + j];
    if (jb->v_in) /* dlib's member v persists between operator() calls of one object (mpc.h:250) */
        for (int t = 0; t < H; ++t)
            for (int j = 0; j < I; ++j) s->v[t][j] = jb->v_in[(long)H * I * k + t * I + j];
    mpc_real u0[2];
    mpc_oracle_step(s, jb->x0 + 2 * k, u0);
    for (int j = 0; j < I; ++j) jb->u0[I * k + j] = u0[j];
    if (jb->v_out)
        for (int t = 0; t < H; ++t)
            for (int j = 0; j < I; ++j) jb->v_out[(long)H * I * k + t * I + j] = s->v[t][j];
    if (jb->controls_out)
        for (int t = 0; t < H; ++t)
            for (int j = 0; j < I; ++j)
                jb->controls_out[(long)H * I * k + t * I + j] = s->controls[t][j];
    if (jb->iters) jb->iters[k] = (int)s->last_iters;
}

static void* worker(void* arg) {
    const job_t* jb = (const job_t*)arg;
    mpc_oracle* s = (mpc_oracle*)malloc(sizeof(mpc_oracle));
    const long chunk = 64; /* interleaved blocks so iteration-count skew spreads over threads */
    for (long b = (long)jb->tid * chunk; b < jb->n; b += (long)jb->nthreads * chunk) {
        const long e = b + chunk < jb->n ? b + chunk : jb->n;
        for (long k = b; k < e; ++k) {
            if (jb->kind == 0) compact_one(jb, k, s);
            else general_one(jb, k, s);
        }
    }
    free(s);
    return NULL;
}

static int run_jobs(job_t* proto, int nthreads) {
    if (proto->H < 1 || proto->H > MPC_ORACLE_MAX_H || (proto->I != 1 && proto->I != 2)) return -1;
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 256) nthreads = 256;
    job_t* jobs = (job_t*)malloc(sizeof(job_t) * nthreads);
    pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * nthreads);
    for (int t = 0; t < nthreads; ++t) {
        jobs[t] = *proto;
        jobs[t].tid = t;
        jobs[t].nthreads = nthreads;
    }
    if (nthreads == 1) {
        worker(&jobs[0]);
    } else {
        for (int t = 0; t < nthreads; ++t) pthread_create(&th[t], NULL, worker, &jobs[t]);
        for (int t = 0; t < nthreads; ++t) pthread_join(th[t], NULL);
    }
    free(jobs);
    free(th);
    return 0;
}

int mpc_oracle_solve_compact(int H, long n, int nthreads, const mpc_real* v, const mpc_real* dy,
                             const mpc_real* dphi, const mpc_real* weights4, double T, double l,
                             const mpc_real* lo2, const mpc_real* hi2, double eps,
                             unsigned long max_iter, unsigned long smo_iters,
                             mpc_real* out_front, mpc_real* out_rear, int* iters) {
    job_t jb;
    memset(&jb, 0, sizeof(jb));
    jb.kind = 0; jb.I = 2; jb.H = H; jb.n = n;
    jb.v = v; jb.dy = dy; jb.dphi = dphi; jb.w = weights4; jb.T = (mpc_real)T; jb.l = (mpc_real)l;
    jb.lo = lo2; jb.hi = hi2; jb.eps = (mpc_real)eps; jb.max_iter = max_iter; jb.smo_iters = smo_iters;
    jb.out_front = out_front; jb.out_rear = out_rear; jb.iters = iters;
    return run_jobs(&jb, nthreads);
}

int mpc_oracle_solve_general(int I, int H, long n, int nthreads, const mpc_real* A,
                             const mpc_real* B, const mpc_real* C, const mpc_real* Q, const mpc_real* R,
                             const mpc_real* lo, const mpc_real* hi, const mpc_real* x0,
                             const mpc_real* targets, const mpc_real* controls_in, double eps,
                             unsigned long max_iter, unsigned long smo_iters, mpc_real* u0,
                             mpc_real* controls_out, int* iters) {
    job_t jb;
    memset(&jb, 0, sizeof(jb));
    jb.kind = 1; jb.I = I; jb.H = H; jb.n = n;
    jb.A = A; jb.B = B; jb.C = C; jb.Q = Q; jb.R = R; jb.lo = lo; jb.hi = hi; jb.x0 = x0;
    jb.targets = targets; jb.controls_in = controls_in; jb.eps = (mpc_real)eps; jb.max_iter = max_iter;
    jb.smo_iters = smo_iters; jb.u0 = u0; jb.controls_out = controls_out; jb.iters = iters;
    return run_jobs(&jb, nthreads);
}

/* The same with the controller's accelerated-gradient memory v (mpc.h:250) carried in and out. */
int mpc_oracle_solve_general_state(int I, int H, long n, int nthreads, const mpc_real* A,
                                   const mpc_real* B, const mpc_real* C, const mpc_real* Q,
                                   const mpc_real* R, const mpc_real* lo, const mpc_real* hi,
                                   const mpc_real* x0, const mpc_real* targets,
                                   const mpc_real* controls_in, const mpc_real* v_in, double eps,
                                   unsigned long max_iter, unsigned long smo_iters, mpc_real* u0,
                                   mpc_real* controls_out, mpc_real* v_out, int* iters) {
    job_t jb;
    memset(&jb, 0, sizeof(jb));
    jb.kind = 1; jb.I = I; jb.H = H; jb.n = n;
    jb.A = A; jb.B = B; jb.C = C; jb.Q = Q; jb.R = R; jb.lo = lo; jb.hi = hi; jb.x0 = x0;
    jb.targets = targets; jb.controls_in = controls_in; jb.v_in = v_in; jb.eps = (mpc_real)eps;
    jb.max_iter = max_iter; jb.smo_iters = smo_iters; jb.u0 = u0; jb.controls_out = controls_out;
    jb.v_out = v_out; jb.iters = iters;
    return run_jobs(&jb, nthreads);
}

/* One solver object called `steps` times: warm start + target shift (mpc.h:229-239), plant
 * update x = A*x + B*u + C as in dlib_files/dlib/test/mpc.cpp:314. */
int mpc_oracle_rollout(int I, int H, int steps, const mpc_real* A, const mpc_real* B,
                       const mpc_real* C, const mpc_real* Q, const mpc_real* R, const mpc_real* lo,
                       const mpc_real* hi, const mpc_real* x0, const mpc_real* targets0,
                       const mpc_real* new_last_targets, double eps, unsigned long max_iter,
                       unsigned long smo_iters, mpc_real* controls_out, mpc_real* states_out,
                       int* iters) {
    if (H < 1 || H > MPC_ORACLE_MAX_H || (I != 1 && I != 2)) return -1;
    mpc_oracle* s = (mpc_oracle*)malloc(sizeof(mpc_oracle));
    mpc_oracle_init(s, I, H, A, B, C, Q, R, lo, hi);
    s->eps = (mpc_real)eps;
    s->max_iter = max_iter;
    s->smo_iters = smo_iters;
    for (int t = 0; t < H; ++t) mpc_oracle_set_target(s, targets0 + 2 * t, t);
    mpc_real x[2] = {x0[0], x0[1]};
    for (int st = 0; st < steps; ++st) {
        if (new_last_targets && st > 0) mpc_oracle_set_target(s, new_last_targets + 2 * st, H - 1);
        mpc_real u[2] = {0, 0};
        mpc_oracle_step(s, x, u);
        for (int j = 0; j < I; ++j) controls_out[I * st + j] = u[j];
        mpc_real xn[2];
        for (int r = 0; r < 2; ++r) {
            mpc_real bu = s->B[r * I + 0] * u[0];
            if (I == 2) bu = bu + s->B[r * I + 1] * u[1];
            xn[r] = ((s->A[r * 2 + 0] * x[0] + s->A[r * 2 + 1] * x[1]) + bu) + s->C[r];
        }
        x[0] = xn[0];
        x[1] = xn[1];
        states_out[2 * st] = x[0];
        states_out[2 * st + 1] = x[1];
        if (iters) iters[st] = (int)s->last_iters;
    }
    free(s);
    return 0;
}
