// C ABI of libtpc_mpc.so (include/tpc_mpc.h): argument validation, host<->device staging, kernel
// family selection.  No solver arithmetic lives here and there is no CPU solve path: every entry
// point ends in a gfx950 kernel launch or fails.
#include "../../include/tpc_mpc.h"

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "mpc_internal.h"

namespace tpc {
// per-horizon launchers, one translation unit each (mpc_lane_inst.hip / mpc_wave_inst.hip)
#define TPC_DECL_H(h)                                                                             \
    hipError_t lane_compact_h##h(int, const CompactArgs&, const Knobs&, const Workspace&, hipStream_t); \
    hipError_t lane_general_h##h(int, int, const GeneralArgs&, const Knobs&, const Workspace&, hipStream_t); \
    hipError_t wave_compact_h##h(int, const CompactArgs&, const Knobs&, const Workspace&, hipStream_t); \
    hipError_t wave_general_h##h(int, int, const GeneralArgs&, const Knobs&, const Workspace&, hipStream_t); \
    int64_t lane_rec_len_h##h(int dtype);
TPC_DECL_H(4) TPC_DECL_H(5) TPC_DECL_H(10) TPC_DECL_H(20) TPC_DECL_H(30) TPC_DECL_H(40)
#undef TPC_DECL_H
}  // namespace tpc

namespace tpc {
struct RolloutStepArgs {
    int64_t n, ld;
    int I, H, step, steps;
    const void *A, *B, *C;
    void* x;
    void* targets;
    const void* controls;
    const void* new_last_targets;
    void* controls_out;
    void* states_out;
    const int32_t* iters_step;
    int32_t* iters_out;
};
hipError_t launch_rollout_step(int dtype, const RolloutStepArgs& a, hipStream_t s);

struct FollowArgs {
    int64_t n, ld;
    int max_points;
    const float *px, *py, *dx, *dy, *vel;
    const int32_t* count;
    const float* car_velocity;
    const float* look_ahead;
    const float *lut_x, *lut_y;
    int lut_n;
    double *v_out, *ysoll_out, *phisoll_out;
    float *target_speed, *target_distance;
};
hipError_t launch_traj_point(const FollowArgs& a, hipStream_t s);
hipError_t launch_follow_post(int64_t n, const float* target_speed, double* front, double* rear, hipStream_t s);
}  // namespace tpc

using namespace tpc;

static const int kHorizons[] = {4, 5, 10, 20, 30, 40};
static std::string g_create_error;

struct tpc_mpc_context {
    int device = 0;
    int cu_count = 0;
    std::string err;
    // device scratch (grown on demand, never shrunk)
    void* ws_state = nullptr;
    int64_t ws_bytes = 0;
    uint32_t* ws_words = nullptr;   // [0] ticket, [1] flags
    // staging for TPC_MPC_HOST batches
    void* stage = nullptr;
    int64_t stage_bytes = 0;
    // rollout working set (state, targets, controller memory, per-step iteration counts)
    void* roll = nullptr;
    int64_t roll_bytes = 0;
    // 64 B of pinned host memory mapped into the device: solve_one's three inputs and two outputs
    // travel through it, so a single solve costs one kernel launch and one sync, no memcpy calls
    void* pin_host = nullptr;
    void* pin_dev = nullptr;
    // queue-order hint for the next batch solve (tpc_mpc_set_work_hint): the caller's device array,
    // or our device copy of a host array
    const int32_t* hint = nullptr;
    int64_t hint_n = 0;
    void* hint_own = nullptr;
    int64_t hint_own_bytes = 0;
    // solve_one has no flags output: it skips the flag word's memset and the kernels' atomicOr
    bool collect_flags = true;
    // optional kernel timing (tpc_mpc_set_profiling)
    bool profiling = false;
    bool ev_valid = false;
    hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
    int last_algo = 0;
};

namespace {

int fail(tpc_mpc_context* h, int code, const std::string& msg) {
    if (h) h->err = msg; else g_create_error = msg;
    return code;
}
int hip_fail(tpc_mpc_context* h, hipError_t e, const char* what) {
    return fail(h, TPC_MPC_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
}
#define HIP_TRY(h, call)                                         \
    do {                                                         \
        hipError_t e__ = (call);                                 \
        if (e__ != hipSuccess) return hip_fail(h, e__, #call);   \
    } while (0)

bool horizon_ok(int H) {
    for (int h : kHorizons) if (h == H) return true;
    return false;
}
size_t esize(int dtype) { return dtype == TPC_MPC_F64 ? 8 : 4; }

int check_common(tpc_mpc_context* h, const tpc_mpc_params* p) {
    if (!h) return fail(nullptr, TPC_MPC_ERR_BAD_ARG, "null handle");
    if (!p) return fail(h, TPC_MPC_ERR_BAD_ARG, "null params");
    if (!horizon_ok(p->horizon))
        return fail(h, TPC_MPC_ERR_BAD_HORIZON, "unsupported horizon " + std::to_string(p->horizon) +
                                                    " (supported: 4 5 10 20 30 40)");
    if (p->dtype != TPC_MPC_F64 && p->dtype != TPC_MPC_F32) return fail(h, TPC_MPC_ERR_BAD_ARG, "bad dtype");
    if (p->algo < TPC_MPC_ALGO_AUTO || p->algo > TPC_MPC_ALGO_LANE) return fail(h, TPC_MPC_ERR_BAD_ARG, "bad algo");
    if (!(p->eps > 0)) return fail(h, TPC_MPC_ERR_BAD_EPS, "eps must be > 0 (mpc.h:202)");
    if (p->max_iter > 0x7fffffffull || p->smo_iters > 0x7fffffffull)
        return fail(h, TPC_MPC_ERR_BAD_ARG, "max_iter / smo_iters must fit in 31 bits");
    return TPC_MPC_OK;
}

// mpc_abstract.h:90-97: min(Q) >= 0, min(R) > 0, min(upper-lower) >= 0
int check_compact_model(tpc_mpc_context* h, const tpc_mpc_params* p) {
    if (!(p->weight_y >= 0) || !(p->weight_phi >= 0))
        return fail(h, TPC_MPC_ERR_BAD_WEIGHTS, "min(Q) >= 0 violated (mpc_abstract.h:90-97)");
    if (!(p->weight_steering_front > 0) || !(p->weight_steering_rear > 0))
        return fail(h, TPC_MPC_ERR_BAD_WEIGHTS, "min(R) > 0 violated (mpc_abstract.h:90-97)");
    for (int j = 0; j < 2; ++j)
        if (!(p->upper[j] >= p->lower[j]))
            return fail(h, TPC_MPC_ERR_BAD_BOUNDS, "upper >= lower violated (mpc_abstract.h:90-97)");
    if (!std::isfinite(p->step_size) || !std::isfinite(p->wheelbase) || p->wheelbase == 0)
        return fail(h, TPC_MPC_ERR_BAD_ARG, "step_size / wheelbase must be finite, wheelbase != 0");
    return TPC_MPC_OK;
}

int ensure(tpc_mpc_context* h, void** buf, int64_t* have, int64_t need) {
    if (need <= *have) return TPC_MPC_OK;
    if (*buf) { hipError_t e = hipFree(*buf); *buf = nullptr; *have = 0; if (e != hipSuccess) return hip_fail(h, e, "hipFree"); }
    const int64_t grow = need + need / 4 + 4096;
    hipError_t e = hipMalloc(buf, (size_t)grow);
    if (e != hipSuccess) return fail(h, TPC_MPC_ERR_ALLOC, std::string("hipMalloc: ") + hipGetErrorString(e));
    *have = grow;
    return TPC_MPC_OK;
}

Knobs knobs_of(const tpc_mpc_params* p) {
    Knobs k;
    k.eps = p->eps;
    k.max_iter = (uint32_t)p->max_iter;
    k.smo_iters = (uint32_t)p->smo_iters;
    return k;
}

// One-shot: the hint set for this handle applies to the next batch solve of the same size only.
const int32_t* take_hint(tpc_mpc_context* h, int64_t n) {
    const int32_t* p = (h->hint && h->hint_n == n) ? h->hint : nullptr;
    h->hint = nullptr;
    h->hint_n = 0;
    return p;
}

// LANE needs enough instances to give every SIMD a full wavefront; below that WAVE's
// one-wavefront-per-instance launch finishes sooner.  Measured crossovers on MI355X
// (scripts/crossover.py, fp64, compact form, of the chip's 65 536 LANE slots): H = 4: 10 000 - 12 000
// instances, H = 10: 16 000 - 24 000, H = 20: 33 000 - 49 000.
// The WAVE kernel maps one decision variable to one lane, so it exists for I*H <= 64 only.
// Returns the kernel family to run, or -1 when WAVE was demanded for a shape it cannot take.
int pick_algo(const tpc_mpc_context* h, int algo, int I, int H, int64_t n) {
    const bool wave_ok = I * H <= kWave;
    if (algo == TPC_MPC_ALGO_WAVE) return wave_ok ? algo : -1;
    if (algo == TPC_MPC_ALGO_LANE) return algo;
    const int64_t lanes = (int64_t)h->cu_count * 4 * kWave;
    const int64_t crossover = H <= 5 ? lanes / 6 : (H <= 10 ? lanes / 3 : lanes * 9 / 16);
    return (n >= crossover || !wave_ok) ? TPC_MPC_ALGO_LANE : TPC_MPC_ALGO_WAVE;
}

int64_t lane_rec_len(int H, int dtype) {
    switch (H) {
#define X(h) case h: return lane_rec_len_h##h(dtype);
        X(4) X(5) X(10) X(20) X(30) X(40)
#undef X
    }
    return 0;
}

hipError_t dispatch_compact(int algo, int H, int dtype, const CompactArgs& a, const Knobs& k,
                            const Workspace& ws, hipStream_t s) {
    switch (H) {
#define X(h) case h: return algo == TPC_MPC_ALGO_LANE ? lane_compact_h##h(dtype, a, k, ws, s) \
                                                       : wave_compact_h##h(dtype, a, k, ws, s);
        X(4) X(5) X(10) X(20) X(30) X(40)
#undef X
    }
    return hipErrorInvalidValue;
}
hipError_t dispatch_general(int algo, int I, int H, int dtype, const GeneralArgs& a, const Knobs& k,
                            const Workspace& ws, hipStream_t s) {
    switch (H) {
#define X(h) case h: return algo == TPC_MPC_ALGO_LANE ? lane_general_h##h(dtype, I, a, k, ws, s) \
                                                       : wave_general_h##h(dtype, I, a, k, ws, s);
        X(4) X(5) X(10) X(20) X(30) X(40)
#undef X
    }
    return hipErrorInvalidValue;
}

int prepare_workspace(tpc_mpc_context* h, int algo, int H, int dtype, int64_t n, Workspace* ws) {
    ws->state = nullptr;
    ws->ticket = h->ws_words;
    ws->stats = (unsigned long long*)(h->ws_words + 4);
    ws->capacity_bytes = 0;
    ws->ev = h->profiling ? h->ev : nullptr;
    h->ev_valid = h->profiling;
    h->last_algo = algo;
    ws->keys = ws->rank = ws->order = nullptr;
    ws->sort_temp = nullptr;
    ws->sort_temp_bytes = 0;
    if (algo == TPC_MPC_ALGO_LANE) {
        // records | keys | rank | order | counting-sort bins
        auto pad = [](int64_t b) { return (b + 255) / 256 * 256; };
        const int64_t rec_b = pad(lane_rec_len(H, dtype) * (int64_t)esize(dtype) * n);
        const int64_t col_b = pad(n * 4);
        const size_t tmp_b = sort_temp_bytes(n);
        int rc = ensure(h, &h->ws_state, &h->ws_bytes, rec_b + 3 * col_b + pad((int64_t)tmp_b));
        if (rc) return rc;
        char* b = (char*)h->ws_state;
        ws->state = b;
        ws->keys = (uint32_t*)(b + rec_b);
        ws->rank = (uint32_t*)(b + rec_b + col_b);
        ws->order = (uint32_t*)(b + rec_b + 2 * col_b);
        ws->sort_temp = b + rec_b + 3 * col_b;
        ws->sort_temp_bytes = tmp_b;
        ws->capacity_bytes = h->ws_bytes;
    }
    return TPC_MPC_OK;
}

int finish_flags(tpc_mpc_context* h, uint32_t* flags_out, hipStream_t s) {
    if (!flags_out) return TPC_MPC_OK;
    uint32_t f = 0;
    HIP_TRY(h, hipMemcpyAsync(&f, h->ws_words + 1, sizeof(f), hipMemcpyDeviceToHost, s));
    HIP_TRY(h, hipStreamSynchronize(s));
    *flags_out = f;
    return TPC_MPC_OK;
}

}  // namespace

extern "C" {

int tpc_mpc_abi_version(void) { return TPC_MPC_ABI_VERSION; }

int tpc_mpc_supported_horizons(int* out, int cap) {
    const int n = (int)(sizeof(kHorizons) / sizeof(kHorizons[0]));
    for (int i = 0; i < n && i < cap && out; ++i) out[i] = kHorizons[i];
    return n;
}

int tpc_mpc_default_params(tpc_mpc_params* p, int horizon) {
    if (!p) return TPC_MPC_ERR_BAD_ARG;
    std::memset(p, 0, sizeof(*p));
    p->horizon = horizon;
    p->dtype = TPC_MPC_F64;
    p->algo = TPC_MPC_ALGO_AUTO;
    p->eps = 0.01;            // mpc.h:104
    p->max_iter = 10000;      // mpc.h:103
    p->smo_iters = 50;        // mpc.h:319
    p->step_size = 0.1;       // src/trajectory_point_follower.cpp:96
    p->wheelbase = 0.21;      // include/trajectory_point_follower.h:47
    p->weight_y = 20;         // src/trajectory_point_follower.cpp:92-95
    p->weight_phi = 7;
    p->weight_steering_front = 0.0005;
    p->weight_steering_rear = 10;
    const double alpha_max = 22 * M_PI / 180;   // src/trajectory_point_follower.cpp:16-18
    p->lower[0] = p->lower[1] = -alpha_max;
    p->upper[0] = p->upper[1] = alpha_max;
    return horizon_ok(horizon) ? TPC_MPC_OK : TPC_MPC_ERR_BAD_HORIZON;
}

int tpc_mpc_create(int device, tpc_mpc_handle* out) {
    if (!out) return fail(nullptr, TPC_MPC_ERR_BAD_ARG, "null out pointer");
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return fail(nullptr, TPC_MPC_ERR_NO_DEVICE,
                    std::string("no HIP device: ") + (e != hipSuccess ? hipGetErrorString(e) : "count == 0"));
    if (device < 0 || device >= count)
        return fail(nullptr, TPC_MPC_ERR_NO_DEVICE, "device index out of range");
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, device);
    if (e != hipSuccess) return fail(nullptr, TPC_MPC_ERR_NO_DEVICE, hipGetErrorString(e));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(nullptr, TPC_MPC_ERR_NO_DEVICE,
                    std::string("device is ") + prop.gcnArchName + ", this library carries gfx950 code only");
    tpc_mpc_context* h = new (std::nothrow) tpc_mpc_context;
    if (!h) return fail(nullptr, TPC_MPC_ERR_ALLOC, "out of host memory");
    h->device = device;
    h->cu_count = prop.multiProcessorCount;
    e = hipSetDevice(device);
    if (e == hipSuccess) e = hipMalloc((void**)&h->ws_words, 64);
    if (e == hipSuccess) e = hipMemset(h->ws_words, 0, 64);
    if (e == hipSuccess) e = hipHostMalloc(&h->pin_host, 64, hipHostMallocMapped);
    if (e == hipSuccess) e = hipHostGetDevicePointer(&h->pin_dev, h->pin_host, 0);
    if (e != hipSuccess) {
        delete h;
        return fail(nullptr, TPC_MPC_ERR_HIP, std::string("create: ") + hipGetErrorString(e));
    }
    *out = h;
    return TPC_MPC_OK;
}

int tpc_mpc_destroy(tpc_mpc_handle h) {
    if (!h) return TPC_MPC_OK;
    (void)hipSetDevice(h->device);
    if (h->ws_state) (void)hipFree(h->ws_state);
    if (h->ws_words) (void)hipFree(h->ws_words);
    if (h->stage) (void)hipFree(h->stage);
    if (h->roll) (void)hipFree(h->roll);
    if (h->hint_own) (void)hipFree(h->hint_own);
    if (h->pin_host) (void)hipHostFree(h->pin_host);
    for (auto& e : h->ev) if (e) (void)hipEventDestroy(e);
    delete h;
    return TPC_MPC_OK;
}

const char* tpc_mpc_last_error(tpc_mpc_handle h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int tpc_mpc_solve_batch_compact(tpc_mpc_handle h, const tpc_mpc_params* p, int64_t n, const void* v,
                                const void* delta_y, const void* delta_phi, void* steering_front,
                                void* steering_rear, int32_t* iters, uint32_t* flags_out, int mem,
                                void* stream) {
    int rc = check_common(h, p);
    if (rc) return rc;
    rc = check_compact_model(h, p);
    if (rc) return rc;
    if (n < 0 || n > 0x7fffffffll) return fail(h, TPC_MPC_ERR_BAD_ARG, "need 0 <= n < 2^31");
    if (mem != TPC_MPC_HOST && mem != TPC_MPC_DEVICE) return fail(h, TPC_MPC_ERR_BAD_ARG, "bad memory kind");
    if (n == 0) { if (flags_out) *flags_out = 0; return TPC_MPC_OK; }
    if (!v || !delta_y || !delta_phi || !steering_front || !steering_rear)
        return fail(h, TPC_MPC_ERR_BAD_ARG, "null batch pointer");
    HIP_TRY(h, hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    const size_t es = esize(p->dtype);
    const int algo = pick_algo(h, p->algo, 2, p->horizon, n);
    if (algo < 0) return fail(h, TPC_MPC_ERR_BAD_HORIZON, "the WAVE kernel needs inputs*horizon <= 64; use LANE or AUTO");

    CompactArgs a;
    std::memset(&a, 0, sizeof(a));
    a.n = n;
    if (mem == TPC_MPC_DEVICE) {
        a.v = v; a.dy = delta_y; a.dphi = delta_phi;
        a.front = steering_front; a.rear = steering_rear; a.iters = iters;
    } else {
        // staging layout: v | dy | dphi | front | rear | iters
        const int64_t col = (int64_t)((n * es + 255) / 256 * 256);
        const int64_t icol = (int64_t)((n * 4 + 255) / 256 * 256);
        rc = ensure(h, &h->stage, &h->stage_bytes, 5 * col + icol);
        if (rc) return rc;
        char* b = (char*)h->stage;
        HIP_TRY(h, hipMemcpyAsync(b, v, n * es, hipMemcpyHostToDevice, s));
        HIP_TRY(h, hipMemcpyAsync(b + col, delta_y, n * es, hipMemcpyHostToDevice, s));
        HIP_TRY(h, hipMemcpyAsync(b + 2 * col, delta_phi, n * es, hipMemcpyHostToDevice, s));
        a.v = b; a.dy = b + col; a.dphi = b + 2 * col;
        a.front = b + 3 * col; a.rear = b + 4 * col;
        a.iters = iters ? (int32_t*)(b + 5 * col) : nullptr;
    }
    a.flags = h->collect_flags ? h->ws_words + 1 : nullptr;
    a.work_hint = take_hint(h, n);
    a.step = p->step_size; a.wheelbase = p->wheelbase;
    a.q[0] = p->weight_y; a.q[1] = p->weight_phi;
    a.r[0] = p->weight_steering_front; a.r[1] = p->weight_steering_rear;
    a.lo[0] = p->lower[0]; a.lo[1] = p->lower[1]; a.hi[0] = p->upper[0]; a.hi[1] = p->upper[1];

    Workspace ws;
    rc = prepare_workspace(h, algo, p->horizon, p->dtype, n, &ws);
    if (rc) return rc;
    if (a.flags) HIP_TRY(h, hipMemsetAsync(h->ws_words + 1, 0, sizeof(uint32_t), s));
    hipError_t e = dispatch_compact(algo, p->horizon, p->dtype, a, knobs_of(p), ws, s);
    if (e != hipSuccess) return hip_fail(h, e, "kernel launch");

    if (mem == TPC_MPC_HOST) {
        HIP_TRY(h, hipMemcpyAsync(steering_front, a.front, n * es, hipMemcpyDeviceToHost, s));
        HIP_TRY(h, hipMemcpyAsync(steering_rear, a.rear, n * es, hipMemcpyDeviceToHost, s));
        if (iters) HIP_TRY(h, hipMemcpyAsync(iters, a.iters, n * 4, hipMemcpyDeviceToHost, s));
        HIP_TRY(h, hipStreamSynchronize(s));
    }
    return finish_flags(h, flags_out, s);
}

int tpc_mpc_solve_one(tpc_mpc_handle h, const tpc_mpc_params* p, double v, double delta_y,
                      double delta_phi, double* steering_front, double* steering_rear) {
    if (!steering_front || !steering_rear) return fail(h, TPC_MPC_ERR_BAD_ARG, "null output pointer");
    if (!p) return fail(h, TPC_MPC_ERR_BAD_ARG, "null params");
    if (!h) return fail(nullptr, TPC_MPC_ERR_BAD_ARG, "null handle");
    tpc_mpc_params q = *p;
    if (q.algo == TPC_MPC_ALGO_AUTO && 2 * q.horizon <= kWave) q.algo = TPC_MPC_ALGO_WAVE;   // one instance: one wavefront
    // inputs and outputs go through the handle's mapped pinned block: [v, dy, dphi, front, rear]
    const size_t es = q.dtype == TPC_MPC_F64 ? 8 : 4;
    char* hp = (char*)h->pin_host;
    char* dp = (char*)h->pin_dev;
    if (q.dtype == TPC_MPC_F64) {
        double* x = (double*)hp;
        x[0] = v; x[1] = delta_y; x[2] = delta_phi;
    } else {
        float* x = (float*)hp;
        x[0] = (float)v; x[1] = (float)delta_y; x[2] = (float)delta_phi;
    }
    // The outputs are pre-set to a NaN with a payload no solve produces; the kernel's two stores
    // into the mapped block are the completion signal the host polls for (a stream synchronise
    // costs 10-20 us more than the solve itself at N = 4).  Bounded: after 2 ms of polling, or if
    // the pattern is still there when the stream has drained, fall back to the stream's verdict.
    const uint64_t sentinel64 = 0x7ff8dead5eedc0deull;
    const uint32_t sentinel32 = 0x7fc5eed1u;
    if (q.dtype == TPC_MPC_F64) { std::memcpy(hp + 3 * es, &sentinel64, 8); std::memcpy(hp + 4 * es, &sentinel64, 8); }
    else { std::memcpy(hp + 3 * es, &sentinel32, 4); std::memcpy(hp + 4 * es, &sentinel32, 4); }
    h->collect_flags = false;
    int rc = tpc_mpc_solve_batch_compact(h, &q, 1, dp, dp + es, dp + 2 * es, dp + 3 * es, dp + 4 * es, nullptr,
                                         nullptr, TPC_MPC_DEVICE, nullptr);
    h->collect_flags = true;
    if (rc) return rc;
    bool done = false;
    {
        const auto t0 = std::chrono::steady_clock::now();
        for (int it = 0; !done; ++it) {
            if (q.dtype == TPC_MPC_F64) {
                const volatile uint64_t* o = (const volatile uint64_t*)(hp + 3 * es);
                done = o[0] != sentinel64 && o[1] != sentinel64;
            } else {
                const volatile uint32_t* o = (const volatile uint32_t*)(hp + 3 * es);
                done = o[0] != sentinel32 && o[1] != sentinel32;
            }
            if (!done && (it & 255) == 255 &&
                std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) break;
        }
    }
    if (!done) HIP_TRY(h, hipStreamSynchronize(nullptr));
    if (q.dtype == TPC_MPC_F64) {
        *steering_front = ((double*)hp)[3];
        *steering_rear = ((double*)hp)[4];
    } else {
        *steering_front = ((float*)hp)[3];
        *steering_rear = ((float*)hp)[4];
    }
    return TPC_MPC_OK;
}

int tpc_mpc_solve_batch_general(tpc_mpc_handle h, const tpc_mpc_params* p,
                                const tpc_mpc_general_io* io, uint32_t* flags_out, int mem,
                                void* stream) {
    int rc = check_common(h, p);
    if (rc) return rc;
    if (!io) return fail(h, TPC_MPC_ERR_BAD_ARG, "null io");
    if (io->inputs != 1 && io->inputs != 2) return fail(h, TPC_MPC_ERR_BAD_ARG, "inputs must be 1 or 2");
    if (io->n < 0 || io->ld < io->n || io->n > 0x7fffffffll) return fail(h, TPC_MPC_ERR_BAD_ARG, "need 0 <= n <= ld, n < 2^31");
    if (mem != TPC_MPC_HOST && mem != TPC_MPC_DEVICE) return fail(h, TPC_MPC_ERR_BAD_ARG, "bad memory kind");
    if (io->n == 0) { if (flags_out) *flags_out = 0; return TPC_MPC_OK; }
    if (!io->A || !io->B || !io->C || !io->Q || !io->R || !io->lower || !io->upper || !io->x0 ||
        !io->targets || !io->u0)
        return fail(h, TPC_MPC_ERR_BAD_ARG, "null batch pointer");
    HIP_TRY(h, hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    const size_t es = esize(p->dtype);
    const int I = io->inputs, H = p->horizon;
    const int64_t n = io->n;
    const int algo = pick_algo(h, p->algo, I, H, n);
    if (algo < 0) return fail(h, TPC_MPC_ERR_BAD_HORIZON, "the WAVE kernel needs inputs*horizon <= 64; use LANE or AUTO");

    GeneralArgs a;
    std::memset(&a, 0, sizeof(a));
    a.n = n;
    a.shift_controls = 1;
    std::vector<std::pair<void*, const void*>> back;   // (host dst, device src) copies after the solve
    std::vector<int64_t> back_bytes;
    if (mem == TPC_MPC_DEVICE) {
        a.ld = io->ld;
        a.A = io->A; a.B = io->B; a.C = io->C; a.Q = io->Q; a.R = io->R; a.lo = io->lower; a.hi = io->upper;
        a.x0 = io->x0; a.targets = io->targets; a.controls = io->controls_inout; a.v = io->v_inout;
        a.u0 = io->u0; a.iters = io->iters;
    } else {
        // Host arrays use leading dimension io->ld; they are staged as they are (ld kept).
        const int64_t ld = io->ld;
        a.ld = ld;
        const int comps[] = {4, 2 * I, 2, 2, I, I, I, 2, 2 * H, H * I, H * I, I};
        const void* src[] = {io->A, io->B, io->C, io->Q, io->R, io->lower, io->upper, io->x0, io->targets,
                             io->controls_inout, io->v_inout, nullptr};
        int64_t off[13];
        int64_t total = 0;
        for (int c = 0; c < 12; ++c) { off[c] = total; total += ((int64_t)comps[c] * ld * es + 255) / 256 * 256; }
        off[12] = total;
        total += (n * 4 + 255) / 256 * 256;
        rc = ensure(h, &h->stage, &h->stage_bytes, total);
        if (rc) return rc;
        char* b = (char*)h->stage;
        for (int c = 0; c < 11; ++c)
            if (src[c]) HIP_TRY(h, hipMemcpyAsync(b + off[c], src[c], (size_t)comps[c] * ld * es, hipMemcpyHostToDevice, s));
        a.A = b + off[0]; a.B = b + off[1]; a.C = b + off[2]; a.Q = b + off[3]; a.R = b + off[4];
        a.lo = b + off[5]; a.hi = b + off[6]; a.x0 = b + off[7]; a.targets = b + off[8];
        a.controls = io->controls_inout ? b + off[9] : nullptr;
        a.v = io->v_inout ? b + off[10] : nullptr;
        a.u0 = b + off[11];
        a.iters = io->iters ? (int32_t*)(b + off[12]) : nullptr;
        back.push_back({io->u0, a.u0}); back_bytes.push_back((int64_t)I * ld * es);
        if (io->controls_inout) { back.push_back({io->controls_inout, a.controls}); back_bytes.push_back((int64_t)H * I * ld * es); }
        if (io->v_inout) { back.push_back({io->v_inout, a.v}); back_bytes.push_back((int64_t)H * I * ld * es); }
        if (io->iters) { back.push_back({io->iters, a.iters}); back_bytes.push_back(n * 4); }
    }
    a.flags = h->ws_words + 1;
    a.work_hint = take_hint(h, n);

    Workspace ws;
    rc = prepare_workspace(h, algo, H, p->dtype, n, &ws);
    if (rc) return rc;
    HIP_TRY(h, hipMemsetAsync(h->ws_words + 1, 0, sizeof(uint32_t), s));
    hipError_t e = dispatch_general(algo, I, H, p->dtype, a, knobs_of(p), ws, s);
    if (e != hipSuccess) return hip_fail(h, e, "kernel launch");
    if (mem == TPC_MPC_HOST) {
        for (size_t i = 0; i < back.size(); ++i)
            HIP_TRY(h, hipMemcpyAsync(back[i].first, back[i].second, (size_t)back_bytes[i], hipMemcpyDeviceToHost, s));
        HIP_TRY(h, hipStreamSynchronize(s));
    }
    return finish_flags(h, flags_out, s);
}

int tpc_mpc_rollout(tpc_mpc_handle h, const tpc_mpc_params* p, const tpc_mpc_general_io* io,
                    int32_t steps, const void* new_last_targets, void* controls_out,
                    void* states_out, int32_t* iters_out, uint32_t* flags_out, int mem,
                    void* stream) {
    int rc = check_common(h, p);
    if (rc) return rc;
    if (!io) return fail(h, TPC_MPC_ERR_BAD_ARG, "null io");
    if (io->inputs != 1 && io->inputs != 2) return fail(h, TPC_MPC_ERR_BAD_ARG, "inputs must be 1 or 2");
    if (io->n < 0 || io->ld < io->n || steps < 0) return fail(h, TPC_MPC_ERR_BAD_ARG, "need 0 <= n <= ld, steps >= 0");
    if (mem != TPC_MPC_HOST && mem != TPC_MPC_DEVICE) return fail(h, TPC_MPC_ERR_BAD_ARG, "bad memory kind");
    if (io->n == 0 || steps == 0) { if (flags_out) *flags_out = 0; return TPC_MPC_OK; }
    if (!io->A || !io->B || !io->C || !io->Q || !io->R || !io->lower || !io->upper || !io->x0 ||
        !io->targets || !controls_out)
        return fail(h, TPC_MPC_ERR_BAD_ARG, "null batch pointer");
    HIP_TRY(h, hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    const size_t es = esize(p->dtype);
    const int I = io->inputs, H = p->horizon;
    const int64_t n = io->n, ld = io->ld;
    const int algo = pick_algo(h, p->algo, I, H, n);
    if (algo < 0) return fail(h, TPC_MPC_ERR_BAD_HORIZON, "the WAVE kernel needs inputs*horizon <= 64; use LANE or AUTO");
    auto pad = [](int64_t b) { return (b + 255) / 256 * 256; };

    // device views of the caller's arrays (staged when they live on the host)
    const int in_comps[] = {4, 2 * I, 2, 2, I, I, I, 2, 2 * H, H * I, H * I, steps * 2};
    const void* in_src[] = {io->A, io->B, io->C, io->Q, io->R, io->lower, io->upper, io->x0, io->targets,
                            io->controls_inout, io->v_inout, new_last_targets};
    const void* dv[12];
    char* out_base = nullptr;
    int64_t o_ctrl = 0, o_states = 0, o_iters = 0;
    if (mem == TPC_MPC_DEVICE) {
        for (int c = 0; c < 12; ++c) dv[c] = in_src[c];
    } else {
        int64_t off[12], total = 0;
        for (int c = 0; c < 12; ++c) { off[c] = total; total += pad((int64_t)in_comps[c] * ld * es); }
        o_ctrl = total; total += pad((int64_t)steps * I * ld * es);
        o_states = total; total += pad((int64_t)steps * 2 * ld * es);
        o_iters = total; total += pad((int64_t)steps * ld * 4);
        rc = ensure(h, &h->stage, &h->stage_bytes, total);
        if (rc) return rc;
        char* b = (char*)h->stage;
        for (int c = 0; c < 12; ++c) {
            dv[c] = in_src[c] ? b + off[c] : nullptr;
            if (in_src[c]) HIP_TRY(h, hipMemcpyAsync(b + off[c], in_src[c], (size_t)in_comps[c] * ld * es, hipMemcpyHostToDevice, s));
        }
        out_base = b;
    }
    // working set: x[2] | targets[2H] | controls[H*I] | v[H*I] | u0[I] | iters[n]
    const int64_t w_x = 0, w_t = w_x + pad(2 * ld * es), w_c = w_t + pad((int64_t)2 * H * ld * es);
    const int64_t w_v = w_c + pad((int64_t)H * I * ld * es), w_u = w_v + pad((int64_t)H * I * ld * es);
    const int64_t w_i = w_u + pad((int64_t)I * ld * es), w_end = w_i + pad(ld * 4);
    rc = ensure(h, &h->roll, &h->roll_bytes, w_end);
    if (rc) return rc;
    char* w = (char*)h->roll;
    HIP_TRY(h, hipMemcpyAsync(w + w_x, dv[7], 2 * ld * es, hipMemcpyDeviceToDevice, s));
    HIP_TRY(h, hipMemcpyAsync(w + w_t, dv[8], (size_t)2 * H * ld * es, hipMemcpyDeviceToDevice, s));
    if (dv[9]) HIP_TRY(h, hipMemcpyAsync(w + w_c, dv[9], (size_t)H * I * ld * es, hipMemcpyDeviceToDevice, s));
    else HIP_TRY(h, hipMemsetAsync(w + w_c, 0, (size_t)H * I * ld * es, s));
    if (dv[10]) HIP_TRY(h, hipMemcpyAsync(w + w_v, dv[10], (size_t)H * I * ld * es, hipMemcpyDeviceToDevice, s));
    else HIP_TRY(h, hipMemsetAsync(w + w_v, 0, (size_t)H * I * ld * es, s));

    GeneralArgs a;
    std::memset(&a, 0, sizeof(a));
    a.n = n; a.ld = ld; a.shift_controls = 1;
    a.A = dv[0]; a.B = dv[1]; a.C = dv[2]; a.Q = dv[3]; a.R = dv[4]; a.lo = dv[5]; a.hi = dv[6];
    a.x0 = w + w_x; a.targets = w + w_t; a.controls = w + w_c; a.v = w + w_v; a.u0 = w + w_u;
    a.iters = (int32_t*)(w + w_i);
    a.flags = h->ws_words + 1;

    RolloutStepArgs r;
    std::memset(&r, 0, sizeof(r));
    r.n = n; r.ld = ld; r.I = I; r.H = H; r.steps = steps;
    r.A = dv[0]; r.B = dv[1]; r.C = dv[2];
    r.x = w + w_x; r.targets = w + w_t; r.controls = w + w_c; r.new_last_targets = dv[11];
    r.controls_out = mem == TPC_MPC_DEVICE ? controls_out : (void*)(out_base + o_ctrl);
    r.states_out = states_out ? (mem == TPC_MPC_DEVICE ? states_out : (void*)(out_base + o_states)) : nullptr;
    r.iters_step = a.iters;
    r.iters_out = iters_out ? (mem == TPC_MPC_DEVICE ? iters_out : (int32_t*)(out_base + o_iters)) : nullptr;

    Workspace ws;
    rc = prepare_workspace(h, algo, H, p->dtype, n, &ws);
    if (rc) return rc;
    HIP_TRY(h, hipMemsetAsync(h->ws_words + 1, 0, sizeof(uint32_t), s));
    const Knobs kn = knobs_of(p);
    for (int st = 0; st < steps; ++st) {
        hipError_t e = dispatch_general(algo, I, H, p->dtype, a, kn, ws, s);
        if (e != hipSuccess) return hip_fail(h, e, "kernel launch");
        r.step = st;
        e = launch_rollout_step(p->dtype, r, s);
        if (e != hipSuccess) return hip_fail(h, e, "rollout step launch");
    }
    // controller state back to the caller
    if (io->controls_inout)
        HIP_TRY(h, hipMemcpyAsync(io->controls_inout, w + w_c, (size_t)H * I * ld * es,
                                  mem == TPC_MPC_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, s));
    if (io->v_inout)
        HIP_TRY(h, hipMemcpyAsync(io->v_inout, w + w_v, (size_t)H * I * ld * es,
                                  mem == TPC_MPC_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, s));
    if (mem == TPC_MPC_HOST) {
        HIP_TRY(h, hipMemcpyAsync(controls_out, out_base + o_ctrl, (size_t)steps * I * ld * es, hipMemcpyDeviceToHost, s));
        if (states_out) HIP_TRY(h, hipMemcpyAsync(states_out, out_base + o_states, (size_t)steps * 2 * ld * es, hipMemcpyDeviceToHost, s));
        if (iters_out) HIP_TRY(h, hipMemcpyAsync(iters_out, out_base + o_iters, (size_t)steps * ld * 4, hipMemcpyDeviceToHost, s));
        HIP_TRY(h, hipStreamSynchronize(s));
    }
    return finish_flags(h, flags_out, s);
}

int tpc_mpc_follow_batch(tpc_mpc_handle h, const tpc_mpc_params* p, const tpc_mpc_trajectories* t,
                         const float* lookup_x, const float* lookup_y, int32_t lookup_n,
                         double* steering_front, double* steering_rear, float* target_speed,
                         float* target_distance, int32_t* iters, uint32_t* flags_out, void* stream) {
    int rc = check_common(h, p);
    if (rc) return rc;
    rc = check_compact_model(h, p);
    if (rc) return rc;
    if (p->dtype != TPC_MPC_F64) return fail(h, TPC_MPC_ERR_BAD_ARG, "follow_batch solves in fp64");
    if (!t) return fail(h, TPC_MPC_ERR_BAD_ARG, "null trajectories");
    if (t->n < 0 || t->ld < t->n || t->n > 0x7fffffffll || t->max_points < 0 || lookup_n < 0)
        return fail(h, TPC_MPC_ERR_BAD_ARG, "need 0 <= n <= ld, n < 2^31, max_points >= 0, lookup_n >= 0");
    if (t->n == 0) { if (flags_out) *flags_out = 0; return TPC_MPC_OK; }
    if (!t->pos_x || !t->pos_y || !t->dir_x || !t->dir_y || !t->velocity || !t->count || !t->car_velocity ||
        !t->look_ahead || !steering_front || !steering_rear || !target_speed || !target_distance ||
        (lookup_n > 0 && (!lookup_x || !lookup_y)))
        return fail(h, TPC_MPC_ERR_BAD_ARG, "null batch pointer");
    HIP_TRY(h, hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    const int64_t n = t->n;
    // v | y_soll | phi_soll (the compact solve's inputs), produced on device
    const int64_t col = (n * 8 + 255) / 256 * 256;
    rc = ensure(h, &h->roll, &h->roll_bytes, 3 * col);
    if (rc) return rc;
    char* b = (char*)h->roll;
    FollowArgs fa;
    fa.n = n; fa.ld = t->ld; fa.max_points = t->max_points;
    fa.px = t->pos_x; fa.py = t->pos_y; fa.dx = t->dir_x; fa.dy = t->dir_y; fa.vel = t->velocity;
    fa.count = t->count; fa.car_velocity = t->car_velocity; fa.look_ahead = t->look_ahead;
    fa.lut_x = lookup_x; fa.lut_y = lookup_y; fa.lut_n = lookup_n;
    fa.v_out = (double*)b; fa.ysoll_out = (double*)(b + col); fa.phisoll_out = (double*)(b + 2 * col);
    fa.target_speed = target_speed; fa.target_distance = target_distance;
    hipError_t e = launch_traj_point(fa, s);
    if (e != hipSuccess) return hip_fail(h, e, "traj_point launch");
    rc = tpc_mpc_solve_batch_compact(h, p, n, fa.v_out, fa.ysoll_out, fa.phisoll_out, steering_front,
                                     steering_rear, iters, nullptr, TPC_MPC_DEVICE, stream);
    if (rc) return rc;
    e = launch_follow_post(n, target_speed, steering_front, steering_rear, s);
    if (e != hipSuccess) return hip_fail(h, e, "follow_post launch");
    return finish_flags(h, flags_out, s);
}

int tpc_mpc_reserve(tpc_mpc_handle h, const tpc_mpc_params* p, int64_t n, int mem) {
    int rc = check_common(h, p);
    if (rc) return rc;
    if (n < 0 || n > 0x7fffffffll) return fail(h, TPC_MPC_ERR_BAD_ARG, "need 0 <= n < 2^31");
    if (mem != TPC_MPC_HOST && mem != TPC_MPC_DEVICE) return fail(h, TPC_MPC_ERR_BAD_ARG, "bad memory kind");
    if (n == 0) return TPC_MPC_OK;
    HIP_TRY(h, hipSetDevice(h->device));
    // the larger of the two kernel families' needs, so that either choice of AUTO is covered
    const bool ev_valid = h->ev_valid;
    const int last_algo = h->last_algo;
    Workspace ws;
    rc = prepare_workspace(h, TPC_MPC_ALGO_LANE, p->horizon, p->dtype, n, &ws);
    h->ev_valid = ev_valid;
    h->last_algo = last_algo;
    if (rc) return rc;
    if (mem == TPC_MPC_HOST) {
        const int64_t col = (int64_t)((n * esize(p->dtype) + 255) / 256 * 256);
        const int64_t icol = (int64_t)((n * 4 + 255) / 256 * 256);
        rc = ensure(h, &h->stage, &h->stage_bytes, 5 * col + icol);
        if (rc) return rc;
    }
    return TPC_MPC_OK;
}

int tpc_mpc_set_work_hint(tpc_mpc_handle h, const int32_t* hint, int64_t n, int mem) {
    if (!h) return fail(nullptr, TPC_MPC_ERR_BAD_ARG, "null handle");
    h->hint = nullptr;
    h->hint_n = 0;
    if (!hint || n == 0) return TPC_MPC_OK;   // cleared
    if (n < 0 || n > 0x7fffffffll) return fail(h, TPC_MPC_ERR_BAD_ARG, "need 0 <= n < 2^31");
    if (mem != TPC_MPC_HOST && mem != TPC_MPC_DEVICE) return fail(h, TPC_MPC_ERR_BAD_ARG, "bad memory kind");
    if (mem == TPC_MPC_HOST) {
        HIP_TRY(h, hipSetDevice(h->device));
        int rc = ensure(h, &h->hint_own, &h->hint_own_bytes, n * 4);
        if (rc) return rc;
        HIP_TRY(h, hipMemcpy(h->hint_own, hint, (size_t)n * 4, hipMemcpyHostToDevice));
        h->hint = (const int32_t*)h->hint_own;
    } else {
        h->hint = hint;
    }
    h->hint_n = n;
    return TPC_MPC_OK;
}

int tpc_mpc_set_profiling(tpc_mpc_handle h, int enable) {
    if (!h) return fail(nullptr, TPC_MPC_ERR_BAD_ARG, "null handle");
    HIP_TRY(h, hipSetDevice(h->device));
    if (enable)
        for (auto& e : h->ev) if (!e) HIP_TRY(h, hipEventCreate(&e));
    h->profiling = enable != 0;
    h->ev_valid = false;
    return TPC_MPC_OK;
}

int tpc_mpc_last_kernel_times(tpc_mpc_handle h, double* first_ms, double* second_ms, int* algo) {
    if (!h) return fail(nullptr, TPC_MPC_ERR_BAD_ARG, "null handle");
    if (!h->ev_valid) return fail(h, TPC_MPC_ERR_BAD_ARG, "no profiled solve on this handle yet");
    HIP_TRY(h, hipEventSynchronize(h->ev[2]));
    float a = 0, b = 0;
    HIP_TRY(h, hipEventElapsedTime(&a, h->ev[0], h->ev[1]));
    HIP_TRY(h, hipEventElapsedTime(&b, h->ev[1], h->ev[2]));
    if (first_ms) *first_ms = a;
    if (second_ms) *second_ms = b;
    if (algo) *algo = h->last_algo;
    return TPC_MPC_OK;
}

int tpc_mpc_last_lane_stats(tpc_mpc_handle h, uint64_t* wave_iterations, uint64_t* refill_blocks) {
    if (!h) return fail(nullptr, TPC_MPC_ERR_BAD_ARG, "null handle");
    HIP_TRY(h, hipSetDevice(h->device));
    unsigned long long st[2] = {0, 0};
    HIP_TRY(h, hipDeviceSynchronize());
    HIP_TRY(h, hipMemcpy(st, h->ws_words + 4, sizeof(st), hipMemcpyDeviceToHost));
    if (wave_iterations) *wave_iterations = st[0];
    if (refill_blocks) *refill_blocks = st[1];
    return TPC_MPC_OK;
}

}  // extern "C"
