/* tpc_mpc.h -- C ABI of the MI355X-native batched MPC-QP solver (libtpc_mpc.so).
 *
 * This is the drop-in boundary for ONE path of lms-org/trajectory_controller: the per-cycle
 * box-constrained MPC QP that TrajectoryPointController::mpcControllerTobi solves through
 * dlib::mpc<2,2,H>::operator().  Every entry point names the reference interface it replaces
 * (paths relative to the reference tree).  The library reproduces dlib's iteration sequence
 * (coordinate descent for `smo_iters` iterations, then accelerated projected gradient, stop when
 * the largest free gradient component is < eps), so in fp64 its outputs equal the reference's.
 *
 * Conventions
 *   - plain C: pointers + sizes, no C++/torch types; no exception crosses this boundary.
 *   - every function returns a tpc_mpc_status (0 = ok); tpc_mpc_last_error() gives text.
 *   - the solve runs on the GPU.  There is no CPU fallback: without a usable gfx950 device
 *     tpc_mpc_create fails with TPC_MPC_ERR_NO_DEVICE.
 *   - a handle is not thread-safe; distinct handles are independent.  The library never keeps a
 *     caller pointer past the call that received it.
 *   - all solves of one handle share its device scratch and therefore never overlap: a solve
 *     submitted to a different stream than the handle's previous one first makes that stream wait
 *     for the previous solve (an event, no host synchronisation).  Batches that should run
 *     concurrently use one handle each.
 *   - batch arrays are structure-of-arrays: component c of instance k lives at base[c*ld + k]
 *     (`ld` >= n is the leading dimension, normally n; a shard of a larger batch passes the
 *     shard's base pointer and the full batch's ld).  Element type is `dtype` (double or float).
 */
#ifndef TPC_MPC_H
#define TPC_MPC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TPC_MPC_ABI_VERSION 5   /* 5 = 4 + new symbols only (the split-named sharded entries); 4 broke 3: tpc_mpc_params.reserved became .options and must be zero-initialised */

typedef struct tpc_mpc_context* tpc_mpc_handle;

typedef enum tpc_mpc_status {
    TPC_MPC_OK = 0,
    TPC_MPC_ERR_BAD_ARG = 1,        /* null pointer, n < 0, ld < n, unknown enum value          */
    TPC_MPC_ERR_BAD_WEIGHTS = 2,    /* min(Q) < 0 or min(R) <= 0   (mpc_abstract.h:90-97)       */
    TPC_MPC_ERR_BAD_BOUNDS = 3,     /* upper < lower               (mpc_abstract.h:90-97)       */
    TPC_MPC_ERR_BAD_HORIZON = 4,    /* horizon outside 1..64, or a kernel family that does not
                                       exist for it was demanded                                */
    TPC_MPC_ERR_BAD_EPS = 5,        /* eps <= 0                    (mpc.h:202 set_epsilon)      */
    TPC_MPC_ERR_NO_DEVICE = 6,      /* no gfx950 GPU / HIP runtime unusable                     */
    TPC_MPC_ERR_HIP = 7,            /* a HIP call failed; text in tpc_mpc_last_error            */
    TPC_MPC_ERR_ALLOC = 8,          /* host or device memory exhausted                          */
    TPC_MPC_ERR_COMM = 9            /* RCCL missing or an RCCL call failed (sharded solves)     */
} tpc_mpc_status;

typedef enum tpc_mpc_dtype { TPC_MPC_F64 = 0, TPC_MPC_F32 = 1 } tpc_mpc_dtype;

/* Where batch arrays live. */
typedef enum tpc_mpc_memory { TPC_MPC_HOST = 0, TPC_MPC_DEVICE = 1 } tpc_mpc_memory;

/* Kernel family.
 *   WAVE : one 64-lane wavefront per instance; lane j owns decision variable j and its row of the
 *          dense (I*H)x(I*H) Hessian (two variables per lane where I*H > 64: I = 2, H <= 64; the
 *          compact form at N = 40 instead builds its gradient from prefix sums over the lanes), the
 *          controls are exchanged by DPP / lane swaps, reductions by wavefront DPP/ballot.  (fp64
 *          batches of more than one instance per SIMD with I*H <= 32 run two instances per wavefront,
 *          one per 32-lane half, and four -- one per 16-lane row -- with I*H <= 16: same arithmetic per
 *          instance, every verdict per group; tpc_mpc_set_option(TPC_MPC_OPT_WAVE_GROUP, 1) keeps
 *          strictly one per wavefront.)  Lowest
 *          latency; used for small and mid-size batches and solve_one.  Needs a specialised horizon
 *          with I*H <= 64 or I = 2.  Agrees with the reference to ~1e-14 (same decisions, different
 *          summation).
 *   LANE : one lane per instance, dlib's O(H) recurrences unrolled in registers, 64 instances
 *          per wavefront with dynamic refill of finished lanes.  Bit-identical to the reference
 *          arithmetic in fp64; the bit-exact family for large batches.  fp64, N = 10, 20, 30, 40 (compact form, and the
 *          general form with or without controller state), batches
 *          that cannot fill the chip that way (below ~98 000 instances at N = 10 / 20, ~60 000 at N = 30 / 40): the
 *          projected-gradient phase runs G = N / 5 lanes per instance (csrc/mpc_lanex.h) -- dlib's two recurrences stay
 *          sequential, handed from lane to lane, everything else of an iteration is shared out; the same IEEE
 *          operations on the same operands, hence the same bits -- 16 384 instances of N = 40: 14.3 ms instead of 49.8,
 *          N = 20: 3.3 instead of 6.9.
 *   LANE_FMA : LANE's layout (one lane per instance, persistent wavefronts, refill queue) with the
 *          arithmetic rebuilt for the hardware instead of for dlib's rounding: controls in unit-box
 *          coordinates (dlib's clamp is the free [0,1] output clamp of the instruction that produces the
 *          value), fused multiply-adds, and the linear term folded into the backward recurrence (nothing
 *          of the model in memory): about half of LANE's instructions per iteration.  Same iteration and
 *          the same decisions as dlib on quantities that differ by rounding: max |du| vs dlib ~2e-13 at
 *          N = 20, ~1e-12 at N = 40, iteration counts identical on the BASELINE workloads (they can differ
 *          only where dlib's largest gradient component comes within rounding of eps).  Compact form
 *          (solve_batch_compact, _mixed, _sharded) with hi > lo finite, specialised horizons.  General
 *          form (solve_batch_general without controls_inout / v_inout, follow_batch) at N = 4, 5, 10, 20:
 *          the same arithmetic in dlib's own coordinates (per-instance bounds may be pinned or infinite)
 *          with the linear term kept, scaled, as dlib keeps it: 1.2 - 1.5x LANE.  Requests it cannot
 *          take (controller state in or out, general form at N = 30 / 40, degenerate compact bounds, other
 *          horizons) run LANE.
 *   GROUP : G lanes per instance (2, 4 or 8; TPC_MPC_OPT_GROUP_LANES pins it), 64 / G instances per wavefront: the
 *          family between WAVE and LANE_FMA, for batches too large for a wavefront each and too small to give every
 *          lane of the chip an instance.  LANE_FMA's arithmetic with the horizon cut into G chunks: each lane runs
 *          its chunk's part of the two recurrences of mpc.h:275-281 in registers, and the chunks are joined by an
 *          exclusive scan over the lanes of the group (the recurrences are affine with a constant matrix, so joining
 *          costs log2 G steps of DPP moves and fused multiply-adds).  Persistent wavefronts, one per SIMD (fp32: two from
 *          the batch size at which that is faster -- a second resident wavefront nearly doubles a SIMD's fp32 issue rate
 *          and adds a tenth to its fp64 one), groups
 *          refilled from LANE_FMA's longest-first queue; the coordinate-descent phase and the records are LANE_FMA's
 *          own.  Compact form, N = 10, 20, 30, 40 (chunks padded where G does not divide N), the requests LANE_FMA
 *          takes; anything else runs LANE_FMA / LANE.  Same statement as LANE_FMA: max |du| vs dlib ~2e-13 at N = 20,
 *          ~1.5e-12 at N = 40, identical iteration counts.  16 384 instances of N = 20: 1.3 ms (WAVE 2.9, LANE_FMA
 *          3.5); of N = 40: 4.8 ms (15.0 / 30.6).
 *          General form (solve_batch_general, rollout, follow_batch*), fp64, N = 10, 20, 30, 40, one or two inputs:
 *          the chunks are joined by scans of 2x2 matrix powers (A is the instance's own: A^(L 2^s) computed once per
 *          refill), controller state in and out (controls_inout / v_inout) included -- behind the bit-exact
 *          coordinate-descent kernel there, and with dlib's own mask as the stop test, because a warm start may lie
 *          outside the box.  16 384 instances of N = 40, two inputs: 7.1 ms (WAVE 51, LANE 61).
 *   AUTO : the fastest family that meets the 1e-6 parity target, and it GUARANTEES that target where a guarantee is
 *          possible.  Family: WAVE below a crossover measured per dtype and horizon, then GROUP with 8, 4, 2 lanes
 *          per instance, then LANE_FMA (csrc/auto_table.h, generated by scripts/measure_crossover.py on a 256-CU
 *          part and scaled by the CU count: at N = 20 in fp64 WAVE below 7 094 instances, GROUP up to 160 530,
 *          LANE_FMA beyond -- in fp32 GROUP up to 321 060; at N = 30 and 40 GROUP is never overtaken, in either dtype and
 *          either form); LANE where none of them takes the request.  Guarantee (fp64, the specialised horizons,
 *          cold starts -- every compact entry point and the general form without controls_inout / v_inout): an
 *          instance that ends on max_iter has not converged, and over thousands of iterations of an ill-conditioned
 *          problem the tolerance families' rounding differences grow (2.3e-5 seen under adversarial parameters), so
 *          such instances are solved ONCE MORE by the bit-exact LANE kernels in the same call and come back with
 *          dlib's bits; every other instance took dlib's decisions on quantities that differ from dlib's by rounding
 *          (<= 1e-9 asserted, ~1e-12 observed).  The second pass costs three empty launches when nothing ended on
 *          the cap; when something did, it lasts max_iter bit-exact iterations (N = 10 / 20 / 30 / 40: G
 *          lanes per instance, 1.4 us per iteration at N = 40 -- 14 ms with dlib's cap of 10 000; BASELINE config 5,
 *          where a tenth of the N = 40 instances end there: 21.5 ms with it, 7 ms without; general form 2.1 us).
 *          A host that prefers the tolerance families' answer for capped instances sets TPC_MPC_PARAM_FAST_CAPPED in
 *          tpc_mpc_params.options; an explicitly demanded family is taken at its word; a host that needs dlib's bits
 *          everywhere asks for LANE. */
typedef enum tpc_mpc_algo {
    TPC_MPC_ALGO_AUTO = 0, TPC_MPC_ALGO_WAVE = 1, TPC_MPC_ALGO_LANE = 2, TPC_MPC_ALGO_LANE_FMA = 3, TPC_MPC_ALGO_GROUP = 4
} tpc_mpc_algo;

/* Non-fatal per-call flags, OR-ed into *flags_out (may be NULL). */
#define TPC_MPC_FLAG_NONFINITE 0x1u  /* an instance had NaN/Inf inputs: it returns the untouched
                                        start point like dlib does (every NaN comparison is
                                        false, mpc.h:298-311) and this flag is raised           */
#define TPC_MPC_FLAG_MAX_ITER  0x2u  /* an instance stopped on max_iter, not on eps             */
/*   (compact form, absurd speeds only -- |step_size * v| beyond 1e4 in fp32, 1e60 in fp64, where dlib's own
 *    intermediates overflow: dlib can stop early on an all-NaN gradient (its 0 * inf products) where these
 *    kernels keep +-inf and run to max_iter with the same controls; outputs agree, the iteration count and
 *    this flag may not.  DESIGN.md section 4.1.)                                                          */
#define TPC_MPC_FLAG_BAD_MODEL 0x4u  /* general form: an instance's Q, R or bounds break dlib's
                                        requires clause (mpc_abstract.h:90-97; min(Q) >= 0,
                                        min(R) > 0, upper >= lower).  dlib asserts (compiled out in
                                        the reference build); here the instance is not solved and
                                        returns its start point at iteration 0                    */

/* tpc_mpc_params.options */
#define TPC_MPC_PARAM_FAST_CAPPED 0x1  /* AUTO only: keep the tolerance family's answer for instances that end on
                                          max_iter instead of solving them once more bit-exactly (see AUTO above) */

/* Solver knobs.  Defaults (tpc_mpc_default_params) are dlib's and the reference module's:
 *   eps 0.01 (mpc.h:104), max_iter 10000 (mpc.h:103), smo_iters 50 (mpc.h:319),
 *   step_size 0.1 (src/trajectory_point_follower.cpp:96), wheelbase 0.21
 *   (include/trajectory_point_follower.h:47), weights 20/7/0.0005/10
 *   (src/trajectory_point_follower.cpp:92-95), bounds +-22 deg (src/...follower.cpp:16-18). */
typedef struct tpc_mpc_params {
    int32_t horizon;          /* H: MPC_HORIZON, include/trajectory_point_follower.h:48          */
    int32_t dtype;            /* tpc_mpc_dtype of the batch arrays and of the arithmetic          */
    int32_t algo;             /* tpc_mpc_algo                                                     */
    int32_t options;          /* TPC_MPC_PARAM_* bits, 0 by default                                */
    double eps;               /* dlib::mpc::set_epsilon        (mpc.h:197-208)                    */
    uint64_t max_iter;        /* dlib::mpc::set_max_iterations (mpc.h:190-195)                    */
    uint64_t smo_iters;       /* mpc.h:319                                                        */
    /* compact ("reference pattern") model, used by solve_one / solve_batch_compact only:         */
    double step_size;         /* mpcParameters.stepSize  T                                        */
    double wheelbase;         /* l                                                                */
    double weight_y, weight_phi, weight_steering_front, weight_steering_rear;
    double lower[2], upper[2];
} tpc_mpc_params;

/* ---- lifetime --------------------------------------------------------------------------------- */

/* Fill *p with the defaults above for horizon H, fp64, algo AUTO. */
int tpc_mpc_default_params(tpc_mpc_params* p, int horizon);

/* Create a solver bound to HIP device `device` (>= 0).  Owns device scratch; no other state.
 * device == TPC_MPC_DEVICE_NONE creates a HOST-ONLY handle: no GPU is looked for or touched, and the one call it serves
 * is tpc_mpc_solve_one -- on the calling thread, in the LANE_FMA family's arithmetic (csrc/tpc_mpc_host.cpp; fp64, the
 * specialised horizons, finite bounds with upper > lower, a CPU with fused multiply-add) -- which is what a host that
 * solves one short-horizon problem per cycle wants ("usable from any single thread without a GPU"): at
 * the reference's N = 4 a core needs ~3 us where the GPU round trip needs ~10.  Every batch entry point returns
 * TPC_MPC_ERR_NO_DEVICE on such a handle.  AUTO's re-solve of capped instances needs the GPU: a host-only handle
 * returns the tolerance answer and raises TPC_MPC_FLAG_MAX_ITER. */
#define TPC_MPC_DEVICE_NONE (-1)
int tpc_mpc_create(int device, tpc_mpc_handle* out);
int tpc_mpc_destroy(tpc_mpc_handle h);

/* Text of the last error on this handle (or of the last failed create when h == NULL). */
const char* tpc_mpc_last_error(tpc_mpc_handle h);

/* dlib::mpc<2,I,H> is a template: any horizon compiles.  Here every horizon 1 <= H <= 64 is accepted;
 * the ones this function lists (writes up to `cap`, returns how many exist) have kernels specialised
 * at compile time (LANE and WAVE), every other one runs a generic kernel with H as a
 * run-time value -- the same arithmetic, the same results (fp64: bit for bit), several times slower. */
int tpc_mpc_supported_horizons(int* out, int cap);
int tpc_mpc_abi_version(void);
/* How this binary was made: ABI version, target, and the LLVM machine scheduler each per-horizon
 * kernel unit was compiled with (csrc/Makefile).  Static text, valid for the process lifetime. */
const char* tpc_mpc_build_info(void);

/* ---- the call being replaced ------------------------------------------------------------------ */

/* Replaces the body of
 *   void TrajectoryPointController::mpcControllerTobi(double v, double delta_y, double delta_phi,
 *                                                      double* steering_front, double* steering_rear)
 * (include/trajectory_point_follower.h:44, src/trajectory_point_follower.cpp:301-389): builds
 * A=[1,Tv;0,1], B=[0,Tv;Tv/l,-Tv/l], C=0, Q, R from `p`, a fresh controller, one target
 * (delta_y, delta_phi) for all steps, x0 = 0, cold start, and returns u0.  `v` is the speed AFTER
 * the module's velocity lookup (src/...follower.cpp:323).  One instance on the GPU (WAVE kernel),
 * served by a resident wavefront that takes requests through a mailbox (see tpc_mpc_set_resident),
 * so the call costs no kernel launch and no synchronisation call -- or, on a host-only handle (TPC_MPC_DEVICE_NONE)
 * and for the horizons TPC_MPC_OPT_HOST_SOLVE_ONE names, on the calling thread (csrc/tpc_mpc_host.cpp). */
int tpc_mpc_solve_one(tpc_mpc_handle h, const tpc_mpc_params* p, double v, double delta_y,
                      double delta_phi, double* steering_front, double* steering_rear);

/* What the last tpc_mpc_solve_one on this handle reported beside its two outputs: the non-fatal flags the
 * batch entries return through flags_out (TPC_MPC_FLAG_NONFINITE: a NaN / Inf speed or target, the call returned
 * dlib's untouched start point (0, 0); TPC_MPC_FLAG_MAX_ITER: the solve was cut off by max_iter) and the
 * iteration count.  The reference's counterpart is the NaN check and log line behind the call
 * (src/trajectory_point_follower.cpp:101-103); dlib itself reports nothing (mpc.h:298-311).  Either pointer
 * may be NULL.  Costs nothing: the resident wavefront stores the word with its outputs.  TPC_MPC_ERR_BAD_ARG
 * before the first solve_one. */
int tpc_mpc_last_flags(tpc_mpc_handle h, uint32_t* flags, int32_t* iters);

/* No reference counterpart.  tpc_mpc_solve_one keeps one wavefront resident on the GPU between
 * calls (fp64, the specialised horizons, algo AUTO or WAVE; other requests take an ordinary launch).
 * The wavefront leaves by itself `idle_timeout_us` after its last request (default 20 000) and is
 * started again by the next call; tpc_mpc_destroy stops it.  While it is resident a device-wide
 * synchronisation elsewhere in the process (hipDeviceSynchronize, hipFree) waits for it, i.e. up to
 * the idle timeout.  idle_timeout_us <= 0 turns the resident mode off: every solve_one is then one
 * kernel launch.
 * Where the CPU can write device memory (hipDeviceAttributeIsLargeBar) the request is written into
 * device memory through the BAR, otherwise into pinned host memory (TPC_MPC_OPT_MAILBOX_HOST forces the
 * latter). */
int tpc_mpc_set_resident(tpc_mpc_handle h, int64_t idle_timeout_us);

/* The same computation for n independent instances in one launch.  Arrays are `p->dtype`,
 * length n, in `mem`.  `iters` (int32, optional) receives each instance's iteration count,
 * `flags_out` (optional) the OR of the per-instance flags.  `stream` is a hipStream_t or NULL;
 * with TPC_MPC_DEVICE memory the call is asynchronous on that stream unless flags_out != NULL
 * (reading the flags synchronises). */
int tpc_mpc_solve_batch_compact(tpc_mpc_handle h, const tpc_mpc_params* p, int64_t n,
                                const void* v, const void* delta_y, const void* delta_phi,
                                void* steering_front, void* steering_rear, int32_t* iters,
                                uint32_t* flags_out, int mem, void* stream);

/* Mixed-horizon batch (BASELINE.json config 5): instance k is solved with horizon horizons[k]
 * (int32, in `mem` like the other arrays); p->horizon is ignored.  The batch is binned by horizon on
 * the device, every bin runs its horizon's kernels, and the outputs come back in the caller's
 * order.  Only the specialised horizons (tpc_mpc_supported_horizons()) can be mixed: any other value
 * anywhere in the batch fails the call with TPC_MPC_ERR_BAD_HORIZON before anything is solved.  Unlike tpc_mpc_solve_batch_compact this call
 * synchronises `stream` once internally (the bin sizes decide the launches). */
int tpc_mpc_solve_batch_compact_mixed(tpc_mpc_handle h, const tpc_mpc_params* p, int64_t n,
                                      const int32_t* horizons, const void* v, const void* delta_y,
                                      const void* delta_phi, void* steering_front, void* steering_rear,
                                      int32_t* iters, uint32_t* flags_out, int mem, void* stream);

/* ---- the general dlib::mpc<2,I,H> surface ------------------------------------------------------- */

/* Per-instance model and state, SoA with leading dimension ld (see header comment):
 *   A[4]  row-major 2x2          B[2*I] row-major 2xI       C[2]  Q[2]  R[I]  lower[I]  upper[I]
 *   x0[2] current_state          targets[H*2]  (step t, state s at component 2*t+s;
 *                                               dlib::mpc::set_target(val,time), mpc.h:142-155)
 *   controls_inout[H*I] optional: the controller's stored controls (mpc.h:363).  As in
 *        operator() (mpc.h:229-239) they are shifted left by one step before the solve (warm
 *        start); on return they hold the solved sequence.  NULL = fresh controller (zeros).
 *   v_inout[H*I] optional: dlib's accelerated-gradient memory (mpc.h:250), which persists across
 *        calls of one controller object.  NULL = zeros in, not written.
 *   u0[I] out: controls[0], what operator() returns.
 * The caller shifts `targets` between calls (mpc.h:236-237) -- tpc_mpc_rollout does it on device. */
typedef struct tpc_mpc_general_io {
    int32_t inputs;           /* I: 1 or 2 */
    int32_t reserved;
    int64_t n, ld;
    const void *A, *B, *C, *Q, *R, *lower, *upper, *x0, *targets;
    void *controls_inout, *v_inout;
    void* u0;
    int32_t* iters;           /* optional */
} tpc_mpc_general_io;

/* Replaces: dlib::mpc<2,I,H> ctor (mpc.h:51-125) + set_target (mpc.h:142-155) + operator()
 * (mpc.h:216-240) for n independent controllers.  Uses p->horizon, dtype, algo, eps, max_iter,
 * smo_iters; the compact-model fields of `p` are ignored. */
int tpc_mpc_solve_batch_general(tpc_mpc_handle h, const tpc_mpc_params* p,
                                const tpc_mpc_general_io* io, uint32_t* flags_out, int mem,
                                void* stream);

/* Closed loop on device: `steps` successive operator() calls per controller with warm start and
 * target shift (mpc.h:229-239), plant update x <- A x + B u + C between calls (the loop of
 * dlib_files/dlib/test/mpc.cpp:301-316).  io->x0 is the initial state and is left untouched;
 * io->targets holds the initial H targets; new_last_targets (optional, SoA [steps*2] per
 * instance) supplies set_last_target() values for steps >= 1 (NULL repeats the last target).
 * controls_out: SoA [steps*I]; states_out (optional): SoA [steps*2]; iters_out (optional, int32):
 * SoA [steps].  io->controls_inout / v_inout (optional) carry the controller state in and out. */
int tpc_mpc_rollout(tpc_mpc_handle h, const tpc_mpc_params* p, const tpc_mpc_general_io* io,
                    int32_t steps, const void* new_last_targets, void* controls_out,
                    void* states_out, int32_t* iters_out, uint32_t* flags_out, int mem,
                    void* stream);

/* ---- batched cycle(): raw trajectories in, CarCommand fields out -------------------------------- */

/* Polylines of n instances, SoA: point i of instance k at base[i*ld + k] (float), `count[k]` points
 * used (<= max_points).  Fields of street_environment::TrajectoryPoint the module reads:
 * position.x/y, directory.x/y, velocity. */
typedef struct tpc_mpc_trajectories {
    int64_t n, ld;
    int32_t max_points, reserved;
    const float *pos_x, *pos_y, *dir_x, *dir_y, *velocity;
    const int32_t* count;
    const float* car_velocity;   /* car->velocity()                                    (follower.cpp:78)  */
    const float* look_ahead;     /* distanceToTrajectoryPoint after the lookup/FOH rule (follower.cpp:66-73) */
} tpc_mpc_trajectories;

/* Replaces, for n independent controllers in DEVICE memory, the tobiMPC branch of cycle():
 * getTrajectoryPoint (src/trajectory_point_follower.cpp:392-443, without the stateful crossing-stop
 * PID of :445-473), the speed clamp and target extraction (:78-85), the velocity lookup (:323;
 * lookup_x/lookup_y of lookup_n entries, ascending x, piecewise linear, clamped; lookup_n = 0 keeps
 * the speed), mpcControllerTobi (:97) and the crossing rule (:277-283: targetSpeed < 0.5 zeroes the
 * steering).  Outputs: steering_front/rear (double), target_speed, target_distance (float;
 * CarCommand::State fields of :114-117).  fp64 solve; uses p's compact-model fields. */
int tpc_mpc_follow_batch(tpc_mpc_handle h, const tpc_mpc_params* p, const tpc_mpc_trajectories* t,
                         const float* lookup_x, const float* lookup_y, int32_t lookup_n,
                         double* steering_front, double* steering_rear, float* target_speed,
                         float* target_distance, int32_t* iters, uint32_t* flags_out, void* stream);

/* The same with ONE TRAJECTORY POINT PER HORIZON STEP: step t of the horizon gets the target
 * (position.y, atan2(directory)) of the polyline point at arc length look_ahead + t * spacing, fed
 * through dlib::mpc::set_target(val, t) semantics (mpc.h:142-155) to the general-form solver; the
 * model, weights and bounds are the compact ones (p).  spacing = step_spacing[k] (float, optional) or,
 * when NULL, |v| * step_size -- the distance driven per step.  Step 0 is tpc_mpc_follow_batch's point,
 * so target_speed / target_distance and the crossing rule are the same; with spacing 0 the whole
 * call equals tpc_mpc_follow_batch.  targets_out (optional, double, SoA [2N][n]) receives the targets
 * used.  The reference module itself sets one target for all steps (src/...follower.cpp:368-371):
 * this is the SURVEY.md 8f-1 extension, its geometry "parity unpinned" like tpc_mpc_follow_batch's. */
int tpc_mpc_follow_batch_horizon(tpc_mpc_handle h, const tpc_mpc_params* p, const tpc_mpc_trajectories* t,
                                 const float* step_spacing, const float* lookup_x, const float* lookup_y,
                                 int32_t lookup_n, double* steering_front, double* steering_rear,
                                 float* target_speed, float* target_distance, double* targets_out,
                                 int32_t* iters, uint32_t* flags_out, void* stream);

/* ---- sharding a batch over the GPUs of a node ---------------------------------------------------- */

/* No reference counterpart (the reference solves one problem per cycle on one CPU thread).  Instances
 * are independent, so a batch shards over GPUs with no exchange during the solve; the one collective
 * is an all-gather of the control outputs over RCCL/xGMI.  One handle per GPU; the handles of a job
 * (one per process, or several in one process) form a communicator the NCCL way: rank 0 obtains an
 * id, the host distributes it, every rank calls tpc_mpc_comm_init_rank.  A handle without a
 * communicator is a world of one.  RCCL is loaded on first use (dlopen); TPC_MPC_ERR_COMM if absent. */
#define TPC_MPC_COMM_ID_BYTES 128
int tpc_mpc_comm_unique_id(void* id, size_t len);
int tpc_mpc_comm_init_rank(tpc_mpc_handle h, const void* id, size_t len, int rank, int world);
int tpc_mpc_comm_destroy(tpc_mpc_handle h);
/* TEST HOOK, not for hosts: a world of one normally needs no communicator and the exchange is skipped.
 * force_communicator != 0 makes the next tpc_mpc_comm_init_rank(h, id, len, 0, 1) build a real one-rank RCCL
 * communicator, so that a one-GPU box runs the library's own ncclAllGather calls; force_ragged != 0 makes the
 * sharded solve take its ragged form (one in-place ncclBroadcast per owner) whatever the sizes. */
int tpc_mpc_comm_test_mode(tpc_mpc_handle h, int force_communicator, int force_ragged);
/* ncclGroupStart / ncclGroupEnd: a single thread that drives several handles brackets its
 * tpc_mpc_comm_init_rank calls, and each round of tpc_mpc_solve_batch_compact_sharded calls, with these. */
int tpc_mpc_group_begin(void);
int tpc_mpc_group_end(void);
/* The contiguous block of a batch of n_total that `rank` of `world` owns (blocks differ by <= 1). */
int tpc_mpc_shard_range(int64_t n_total, int rank, int world, int64_t* first, int64_t* count);
/* How a batch is split over the ranks (ABI 5).  BLOCK: rank r owns one contiguous block (tpc_mpc_shard_range).
 * INTERLEAVED: rank r owns the instances i = r (mod world).  The iteration count of an instance is a function of its
 * speed (dlib's lambda, mpc.h:116-123), so a batch that arrives SORTED by speed gives one rank of a block split all the
 * 10 000-iteration instances; the interleaved split deals them round.  For iid inputs the two cost the same.
 * tpc_mpc_shard_map: element j of rank's shard is instance first + j * stride, j < count. */
#define TPC_MPC_SPLIT_BLOCK 0
#define TPC_MPC_SPLIT_INTERLEAVED 1
int tpc_mpc_shard_map(int64_t n_total, int rank, int world, int split, int64_t* first, int64_t* count, int64_t* stride);

/* tpc_mpc_solve_batch_compact for this rank's block of a batch of n_total, then the all-gather:
 * v/delta_y/delta_phi_shard hold the block's `count` instances, steering_front_all / _rear_all are
 * FULL-size arrays [n_total]; the block is solved straight into its slot [first, first+count) and the
 * slots are exchanged in place (ncclAllGather when n_total divides evenly, otherwise one in-place
 * ncclBroadcast per owner inside one group), so on return -- in stream order -- every GPU holds all
 * control outputs.  DEVICE memory only; iters_shard [count] and flags_out cover this rank's block. */
int tpc_mpc_solve_batch_compact_sharded(tpc_mpc_handle h, const tpc_mpc_params* p, int64_t n_total,
                                        const void* v_shard, const void* delta_y_shard,
                                        const void* delta_phi_shard, void* steering_front_all,
                                        void* steering_rear_all, int32_t* iters_shard,
                                        uint32_t* flags_out, void* stream);

/* The same with the split named (ABI 5).  TPC_MPC_SPLIT_BLOCK: exactly the call above.  TPC_MPC_SPLIT_INTERLEAVED: the shard
 * arrays hold the rank's `count` instances compacted (element j = instance rank + j * world: the host scatters with a
 * stride); the shard is solved into the rank's slot of a [world][ceil(n_total / world)] staging array owned by the
 * handle, the slots are all-gathered there (equal sizes by construction: one ncclAllGather per output, no ragged form)
 * and one kernel per output writes them back in INSTANCE order, so the full-size outputs look the same under either
 * split -- bit for bit within one kernel family.  iters_shard [count] stays in shard order. */
int tpc_mpc_solve_batch_compact_sharded_split(tpc_mpc_handle h, const tpc_mpc_params* p, int64_t n_total, int split,
                                              const void* v_shard, const void* delta_y_shard,
                                              const void* delta_phi_shard, void* steering_front_all,
                                              void* steering_rear_all, int32_t* iters_shard,
                                              uint32_t* flags_out, void* stream);

/* No reference counterpart (the reference drives one dlib::mpc object on one device: src/trajectory_point_follower.cpp:366).
 * The general form sharded the same way: io_all describes the FULL batch (n = n_total, ld >= n_total, DEVICE memory;
 * every rank passes the same shapes).  The rank solves columns [first, first + count) of those arrays in place -- only
 * that block of the inputs has to be filled on this rank -- and the I rows of u0 are exchanged in place as above, so on
 * return every GPU holds all of u0.  controls_inout / v_inout / iters, when given, are read and written for this
 * rank's block only and not exchanged; flags_out covers this rank's block. */
int tpc_mpc_solve_batch_general_sharded(tpc_mpc_handle h, const tpc_mpc_params* p,
                                        const tpc_mpc_general_io* io_all, uint32_t* flags_out, void* stream);

/* No reference counterpart.  The exchange of the two calls above by itself, for every OTHER entry point a host wants
 * to shard (tpc_mpc_solve_batch_compact_mixed, tpc_mpc_follow_batch*, tpc_mpc_rollout, ...): each rank calls that entry
 * for its block -- tpc_mpc_shard_range says which -- with output pointers offset to the block's slot [first, first + count)
 * of FULL-size arrays, then this.  rows[i] (i < n_rows) is the DEVICE base of one full-size output row of n_total
 * elements of elem_bytes (4 or 8) each; on return -- in stream order -- every GPU holds every rank's slot of every row
 * (ncclAllGather when n_total divides evenly, otherwise one in-place ncclBroadcast per owner; one RCCL group).  A handle
 * without a communicator: a no-op. */
int tpc_mpc_gather_shards(tpc_mpc_handle h, int64_t n_total, void* const* rows, int n_rows, int elem_bytes, void* stream);
/* The same with the split named (ABI 5).  TPC_MPC_SPLIT_INTERLEAVED: on entry the FIRST `count` elements of each row are
 * this rank's shard (element j = instance rank + j * world -- what an entry called on the compacted shard leaves when its
 * output pointer is the row's base); on return the row holds all n_total elements in instance order. */
int tpc_mpc_gather_shards_split(tpc_mpc_handle h, int64_t n_total, int split, void* const* rows, int n_rows, int elem_bytes,
                                void* stream);

/* ---- memory ------------------------------------------------------------------------------------ */

/* No reference counterpart.  A handle grows its device scratch on demand, and growing frees and
 * allocates device memory, which synchronises the whole device; a host that keeps several batches
 * in flight (one handle and one stream per batch) calls this once per handle up front for the
 * largest compact batch it will solve (mem: where the batch arrays will live), so that no solve
 * allocates. */
int tpc_mpc_reserve(tpc_mpc_handle h, const tpc_mpc_params* p, int64_t n, int mem);

/* ---- library options -------------------------------------------------------------------------- */

/* No reference counterpart: switches of this implementation that a host may want to pin (A/B measurements,
 * boards without a large BAR).  They belong to the handle -- the library reads nothing from the environment.
 *   TPC_MPC_OPT_WAVE_GROUP   instances per wavefront of the fp64 WAVE kernels: 0 (default) = as many as fit and
 *                            pay (4 with inputs*horizon <= 16, 2 with <= 32, from one wavefront per SIMD on),
 *                            1 = strictly one, 2 / 4 = pairs / fours where they fit.  Same arithmetic per instance.
 *   TPC_MPC_OPT_MAILBOX_HOST 1 = tpc_mpc_solve_one's request lines live in pinned host memory even where the
 *                            CPU could write device memory through the BAR (0, default: device memory where
 *                            hipDeviceAttributeIsLargeBar says so).  Restarts the resident wavefront (the idle
 *                            timeout and "resident off" of tpc_mpc_set_resident are kept).
 *   TPC_MPC_OPT_GROUP_LANES  lanes per instance of the GROUP family: 0 (default) = the measured best for the batch
 *                            size (csrc/auto_table.h), 2 / 4 / 8 where that size is built for the horizon (N = 10:
 *                            2, 4; N = 20, 30, 40: 2, 4, 8).
 *   TPC_MPC_OPT_HOST_SOLVE_ONE  tpc_mpc_solve_one on the CALLING THREAD for horizons up to this value (0, default:
 *                            never): the host path of csrc/tpc_mpc_host.cpp, as on a handle created with
 *                            TPC_MPC_DEVICE_NONE.  For a module that solves one short-horizon problem per cycle --
 *                            the reference's own N = 4 -- set it to 10: a core needs ~4 us at N = 4 (28 at N = 10) where the
 *                            round trip to the resident wavefront needs ~10 (30); from N = 20 on the GPU is faster
 *                            (INTEGRATION.md section 1). */
typedef enum tpc_mpc_option { TPC_MPC_OPT_WAVE_GROUP = 1, TPC_MPC_OPT_MAILBOX_HOST = 2, TPC_MPC_OPT_GROUP_LANES = 3,
                              TPC_MPC_OPT_HOST_SOLVE_ONE = 4 } tpc_mpc_option;
int tpc_mpc_set_option(tpc_mpc_handle h, int option, int64_t value);

/* ---- measurement ------------------------------------------------------------------------------ */

/* No reference counterpart (the reference times nothing on this branch; its only timers are
 * logger.time("mikMPC") around the other back-end, src/trajectory_point_follower.cpp:134,213).
 * With profiling on, every solve records HIP events on its launch stream around each kernel;
 * tpc_mpc_last_kernel_times waits for the last solve and returns the two kernel durations in ms
 * (LANE: coordinate-descent kernel, projected-gradient kernel; WAVE: the one kernel, 0) and the
 * tpc_mpc_algo that ran. */
int tpc_mpc_set_profiling(tpc_mpc_handle h, int enable);
int tpc_mpc_last_kernel_times(tpc_mpc_handle h, double* first_ms, double* second_ms, int* algo);

/* Occupancy statistics of the last LANE solve (waits for that solve): loop iterations executed
 * by all persistent wavefronts of the projected-gradient kernel, and refill blocks executed.
 * lane utilisation = sum of per-instance PG iterations / (64 * wave_iterations). */
int tpc_mpc_last_lane_stats(tpc_mpc_handle h, uint64_t* wave_iterations, uint64_t* refill_blocks);

#ifdef __cplusplus
}
#endif
#endif /* TPC_MPC_H */
