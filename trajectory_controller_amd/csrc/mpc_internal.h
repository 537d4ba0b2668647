// Internal launch interface between the C ABI (tpc_mpc_api.cpp) and the gfx950 kernels.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tpc {

constexpr int kWave = 64;   // CDNA4 wavefront width

// Solver knobs shared by every kernel (fp64 on the host side; cast to T in the kernels).
struct Knobs {
    double eps;
    uint32_t max_iter;
    uint32_t smo_iters;
};

// Compact ("reference pattern") batch: model built from v per instance
// (reference: src/trajectory_point_follower.cpp:326-333, :359-363).
struct CompactArgs {
    int64_t n;
    const void *v, *dy, *dphi;     // [n]
    void *front, *rear;            // [n]
    int32_t* iters;                // [n] or null
    uint32_t* flags;               // device word, OR-ed with TPC_MPC_FLAG_*, or null
    const int32_t* work_hint;      // [n] or null: caller's iteration-count estimate (LANE queue order only)
    double step, wheelbase;        // T, l
    double q[2], r[2], lo[2], hi[2];
};

// One compact instance whose inputs are already in registers (the resident solve_one wavefront,
// tpc_mpc_one.hip): no input arrays; the two outputs are stored system-wide into out[0], out[1].
struct OneArgs {
    double v, dy, dphi;
    double step, wheelbase;
    double q[2], r[2], lo[2], hi[2];
    uint64_t* out;                         // front, rear (8-byte stores visible to the host)
    uint64_t* info;                        // (TPC_MPC_FLAG_* << 32) | iteration count, stored like the outputs
    static constexpr int32_t* iters = nullptr;
    static constexpr uint32_t* flags = nullptr;
};

// General batch: SoA with leading dimension ld, component c of instance k at base[c*ld + k].
struct GeneralArgs {
    int64_t n, ld;
    const void *A, *B, *C, *Q, *R, *lo, *hi, *x0, *targets;
    void *controls, *v;            // in/out, may be null
    void* u0;
    int32_t* iters;
    uint32_t* flags;
    const int32_t* work_hint;      // as in CompactArgs
    int shift_controls;            // 1: apply operator()'s warm-start shift (mpc.h:231-232)
};

// Device scratch owned by the handle (LANE kernels: state handed from the coordinate-descent
// phase to the projected-gradient phase, and the work-queue ticket).
struct Workspace {
    void* state;        // per-instance records, LaneRec<T,H>::kLen elements each
    uint32_t* ticket;   // LANE: 1 word; WAVE queue: kQueueTickets words, kQueueTicketStride apart
    unsigned long long* stats;   // [3]: PG wave-iterations, refill blocks, exact-stop-test flag (zeroed per launch)
    int64_t capacity_bytes;
    // longest-first queue (mpc_sort.hip): keys written by the CD kernel, ordered into order[]
    uint32_t *keys, *rank, *order;
    void* sort_temp;
    size_t sort_temp_bytes;
    // optional profiling: events recorded on the launch stream around each kernel
    // (ev[0] .. ev[1] first kernel, ev[1] .. ev[2] second kernel); null when profiling is off
    hipEvent_t* ev;
    int wave_group = 0;   // TPC_MPC_OPT_WAVE_GROUP: 0 auto, 1 / 2 / 4 instances per wavefront
    int group_lanes = 0;  // GROUP: lanes per instance (2 / 4 / 8) for this horizon
    int max_waves = 0;    // GROUP: size of the persistent grid (0: one wavefront per SIMD of the device, or per group_pair)
    bool group_pair = false;   // GROUP: as many wavefronts per SIMD as the kernel is built for (fp32: two) -- the batch is past auto_table.h's pair_from
    int64_t lanex_below = -1;   // LANE, compact form, fp64: batches below this take the G-lanes-per-instance kernel (mpc_lanex.h); -1: the measured default, 0: never
    int cu_count = 0;     // of the device (scales measured crossovers); 0: unknown, take 256
};

// WAVE work queue (mpc_wave.h): the dynamic part of the queue is dealt out through kQueueTickets counters on
// cache lines of their own -- returning atomics on one address complete at ~30 M/s on this chip, whatever the
// number of wavefronts asking
constexpr int kQueueTickets = 16;
constexpr int kQueueTicketStride = 64;   // in words: 256 B
constexpr int64_t kQueueTicketBytes = (int64_t)kQueueTickets * kQueueTicketStride * 4;
constexpr int64_t kWaveQueueMaxInstances = 32768;   // what one wave_order_kernel workgroup sorts (mpc_wave.h)

// mpc_generic.hip: the any-horizon fallback (run-time H, per-instance arrays in a global workspace)
constexpr int kAlgoMixed = 101;     // last_algo after a mixed-horizon batch whose bins ran on child handles
constexpr int kAlgoGeneric = 100;   // internal kernel-family code next to TPC_MPC_ALGO_WAVE / _LANE / _LANE_FMA
constexpr int kMaxHorizon = 64;
int64_t generic_scratch_bytes(int H, int dtype, int64_t n);
hipError_t generic_compact(int dtype, int H, const CompactArgs& a, const Knobs& k, void* scratch, hipStream_t s);
hipError_t generic_general(int dtype, int I, int H, const GeneralArgs& a, const Knobs& k, void* scratch, hipStream_t s);

// mpc_sort.hip
size_t sort_temp_bytes(int64_t n);
hipError_t order_begin(void* temp, hipStream_t s);   // zero the bins the key producer counts into
hipError_t presolve_merge(const uint32_t* queue, const uint32_t* queue_len, int64_t n, const void* sf, const void* sr,
                          const int32_t* si, void* f, void* r, int32_t* it, uint32_t limit, hipStream_t s);   // AUTO's presolve (tpc_mpc_api.cpp)
hipError_t presolve_merge_rows(const uint32_t* queue, const uint32_t* queue_len, int64_t n, const void* su, const int32_t* si,
                               void* u, int32_t* it, int rows, int64_t ld, uint32_t limit, hipStream_t s);
const uint32_t* order_queue_len(const void* temp);   // device word: queue entries above the lowest 128 bins
hipError_t order_finish(const uint32_t* keys, const uint32_t* rank, uint32_t* order, int64_t n, void* temp,
                        hipStream_t s);

// mpc_rollout.hip: what the caller of dlib::mpc does between two operator() calls of a closed loop.
// Three leading dimensions: the handle's working set (model copy, state, targets, controls), the
// caller's output arrays, and the caller's new_last_targets.
struct RolloutStepArgs {
    int64_t n, ld, ld_out, ld_nlt;
    int I, H, step, steps;
    const void *A, *B, *C;        // model (SoA, ld)
    void* x;                      // [2] current state, updated in place (ld)
    void* targets;                // [2H] shifted in place (ld)
    const void* controls;         // [H*I] solved sequence of this step (ld)
    const void* new_last_targets; // [steps*2] or null (ld_nlt)
    void* controls_out;           // [steps*I] (ld_out)
    void* states_out;             // [steps*2] or null (ld_out)
    const int32_t* iters_step;    // [n] or null
    int32_t* iters_out;           // [steps] or null (ld_out)
};
hipError_t launch_rollout_step(int dtype, const RolloutStepArgs& a, hipStream_t s);

// mpc_follow.hip: batched front and back ends of the tobiMPC branch of cycle().
struct FollowArgs {
    int64_t n, ld;
    int max_points;
    const float *px, *py, *dx, *dy, *vel;   // [max_points][ld]
    const int32_t* count;                   // [n]
    const float* car_velocity;              // [n]
    const float* look_ahead;                // [n]
    const float *lut_x, *lut_y;             // velocity lookup (ascending x), may be null
    int lut_n;
    double *v_out, *ysoll_out, *phisoll_out;   // [n] -> inputs of the compact solve
    float *target_speed, *target_distance;     // [n]
};
hipError_t launch_traj_point(const FollowArgs& a, hipStream_t s);
// ... one trajectory point per horizon step (tpc_mpc_follow_batch_horizon): the kernel writes a whole
// general-form batch -- the compact model spelled out as A, B, C, Q, R, bounds, x0 = 0 -- plus
// targets[2H], all SoA with leading dimension ldw, for the general-form solver to consume.
struct FollowHorizonArgs {
    int H;
    int64_t ldw;
    const float* step_spacing;     // [n] arc length between consecutive horizon targets, or null: |v|*T
    double step, wheelbase;        // T, l
    double q[2], r[2], lo[2], hi[2];
    double *A, *B, *C, *Q, *R, *lo_out, *hi_out, *x0, *targets;
    double* targets_copy;          // optional [2H][ld_copy] copy of the targets for the caller
    int64_t ld_copy;
};
hipError_t launch_traj_horizon(const FollowArgs& a, const FollowHorizonArgs& f, hipStream_t s);
hipError_t launch_follow_post(int64_t n, const float* target_speed, double* front, double* rear, hipStream_t s);

}  // namespace tpc
