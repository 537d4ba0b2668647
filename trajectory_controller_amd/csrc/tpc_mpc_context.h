// The handle behind tpc_mpc_handle and the helpers every C-ABI translation unit shares
// (tpc_mpc_api.cpp, tpc_mpc_mixed.hip, tpc_mpc_comm.cpp, tpc_mpc_one.hip).  Internal: nothing here is
// visible through include/tpc_mpc.h.
#pragma once

#include "../../include/tpc_mpc.h"

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <exception>
#include <new>

#include "mpc_internal.h"

namespace tpc {
struct OneShot;   // a handle's device-side state for `device` (what tpc_mpc_create makes); the caller has checked the device
int context_new(int device, int cu_count, tpc_mpc_context** out);
// tpc_mpc_one.hip: the resident single-solve kernel's host side
struct Comm;      // tpc_mpc_comm.cpp: RCCL communicator of a sharded solve
}  // namespace tpc

// Error text lives in fixed buffers: reporting an error never allocates, so it cannot throw.
constexpr size_t kTpcErrLen = 512;

struct tpc_mpc_context {
    int device = 0;
    int cu_count = 0;                // what AUTO's crossovers are scaled to: the device's, unless tpc_mpc_x_set_group_share overrides it
    int device_cu_count = 0;         // the device's own (cu_count == 0 in that call restores it)
    char err[kTpcErrLen] = "";
    // device scratch (grown on demand, never shrunk)
    void* ws_state = nullptr;
    int64_t ws_bytes = 0;
    uint32_t* ws_words = nullptr;   // [0] ticket, [1] flags, [4..] lane statistics; [16..23] the same for the re-solve of capped
                                    // instances: [16] ticket, [17] queue length, [18..23] statistics (one memset)
    // staging for TPC_MPC_HOST batches
    void* stage = nullptr;
    int64_t stage_bytes = 0;
    // working set of tpc_mpc_rollout / tpc_mpc_follow_batch* (model copy, state, targets, controller memory)
    void* roll = nullptr;
    int64_t roll_bytes = 0;
    // iteration counts of a tolerance family's pass when the caller asked for none (AUTO re-solves what ended on the cap)
    void* cap_iters = nullptr;
    int64_t cap_iters_bytes = 0;
    // AUTO's presolve (tpc_mpc_api.cpp, presolve_begin): the instances predicted to end on the iteration cap are solved
    // bit-exactly on a stream of the handle's own BESIDE the tolerance family's pass: LANE scratch + side outputs, stream, events
    void* pre = nullptr;
    int64_t pre_bytes = 0;
    hipStream_t pre_stream = nullptr;
    hipEvent_t pre_fork = nullptr, pre_done = nullptr;
    bool pre_busy = false;           // a presolve is in flight on pre_stream (one at a time per handle)
    int pre_group_waves = 0;         // ... and this many wavefronts are what it leaves to every other persistent grid of the handle
    // mixed-horizon batches: bin-contiguous copies of the inputs and outputs, permutation, counters
    void* mix = nullptr;
    int64_t mix_bytes = 0;
    // pinned host memory mapped into the device: solve_one's mailbox (tpc_mpc_one.hip)
    void* pin_host = nullptr;
    void* pin_dev = nullptr;
    tpc::OneShot* one = nullptr;
    // queue-order hint for the next batch solve (tpc_mpc_set_work_hint): always the handle's own copy
    const int32_t* hint = nullptr;
    int64_t hint_n = 0;
    void* hint_own = nullptr;
    int64_t hint_own_bytes = 0;
    // internal solves without a flags output skip the flag word's atomicOr
    bool collect_flags = true;
    // what the last tpc_mpc_solve_one reported (tpc_mpc_last_flags)
    uint32_t one_flags = 0;
    int32_t one_iters = 0;
    bool one_valid = false;
    // optional kernel timing (tpc_mpc_set_profiling)
    bool profiling = false;
    bool ev_valid = false;
    hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
    int last_algo = 0;
    // All solves of a handle share its scratch, so they must not overlap.  The handle remembers the
    // stream of its last solve and an event recorded behind it; a solve submitted to a different
    // stream first makes that stream wait for the event (tpc_mpc_api.cpp, StreamOrder).
    hipEvent_t done_ev = nullptr;
    hipStream_t last_stream = nullptr;
    bool have_last = false;
    // sharded solves (tpc_mpc_comm.cpp); null = a world of one
    tpc::Comm* comm = nullptr;
    void* gather = nullptr;          // [world][cap] staging of the interleaved split's exchange
    int64_t gather_bytes = 0;
    bool comm_test_force = false, comm_test_ragged = false;   // tpc_mpc_comm_test_mode
    // mixed-horizon batches run their bins concurrently: one child handle (scratch of its own) and one stream
    // per bin, forked from and joined back into the caller's stream (tpc_mpc_mixed.hip)
    static constexpr int kMaxKids = 8;
    tpc_mpc_context* kids[kMaxKids] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    hipStream_t kid_stream[kMaxKids] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    hipEvent_t kid_done[kMaxKids] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    hipEvent_t fork_ev = nullptr;
    // tpc_mpc_set_option
    int opt_wave_group = 0;          // 0 auto, 1 / 2 / 4 instances per wavefront (fp64 WAVE)
    bool opt_mailbox_host = false;   // solve_one's request lines in pinned host memory
    int64_t one_idle_us = 20000;     // tpc_mpc_set_resident: idle timeout of the resident wavefront; <= 0 = resident mode off
    int opt_group_lanes = 0;         // GROUP: 0 auto, 2 / 4 / 8 lanes per instance
    int opt_host_horizon = 0;        // tpc_mpc_solve_one on the calling thread for horizons up to this (0: never)
    bool host_only = false;          // created with TPC_MPC_DEVICE_NONE: no HIP state at all
    int64_t opt_lanex_below = -1;    // tpc_mpc_x_set_lanex_below: -1 the measured default, 0 never, else the batch size below which LANE runs mpc_lanex.h
    int max_waves = 0;               // persistent-grid limit of this handle's GROUP solves (child handles of a mixed batch: their share of the chip)
};

namespace tpc {

extern thread_local char g_create_error[kTpcErrLen];   // last failed call without a handle, per thread

__attribute__((format(printf, 3, 4))) inline int fail(tpc_mpc_context* h, int code, const char* fmt, ...) noexcept {
    va_list ap;
    va_start(ap, fmt);
    std::vsnprintf(h ? h->err : g_create_error, kTpcErrLen, fmt, ap);
    va_end(ap);
    return code;
}
inline int hip_fail(tpc_mpc_context* h, hipError_t e, const char* what) noexcept {
    return fail(h, TPC_MPC_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
}
#define HIP_TRY(h, call)                                              \
    do {                                                              \
        hipError_t e__ = (call);                                      \
        if (e__ != hipSuccess) return ::tpc::hip_fail(h, e__, #call); \
    } while (0)

// Every extern "C" body runs inside this: whatever a C++ runtime call might throw (std::bad_alloc
// from the HIP runtime's own containers included) is turned into a status code here and never
// crosses the C boundary (include/tpc_mpc.h, Conventions).
template <class F> int guarded(tpc_mpc_context* h, F&& body) noexcept {
    try {
        return body();
    } catch (const std::bad_alloc&) {
        return fail(h, TPC_MPC_ERR_ALLOC, "out of host memory");
    } catch (const std::exception& e) {
        return fail(h, TPC_MPC_ERR_HIP, "unexpected exception: %s", e.what());
    } catch (...) {
        return fail(h, TPC_MPC_ERR_HIP, "unexpected exception");
    }
}

inline size_t esize(int dtype) { return dtype == TPC_MPC_F64 ? 8 : 4; }
inline int64_t pad256(int64_t bytes) { return (bytes + 255) / 256 * 256; }

// Grow-only device buffer.  Growing frees and allocates, which synchronises the device.
inline int ensure(tpc_mpc_context* h, void** buf, int64_t* have, int64_t need) {
    if (need <= *have) return TPC_MPC_OK;
    if (*buf) {
        hipError_t e = hipFree(*buf);
        *buf = nullptr;
        *have = 0;
        if (e != hipSuccess) return hip_fail(h, e, "hipFree");
    }
    const int64_t grow = need + need / 4 + 4096;
    hipError_t e = hipMalloc(buf, (size_t)grow);
    if (e != hipSuccess) {
        *buf = nullptr;
        (void)hipGetLastError();   // the failed allocation must not poison the next launch check
        return fail(h, TPC_MPC_ERR_ALLOC, "hipMalloc of %lld bytes: %s", (long long)grow, hipGetErrorString(e));
    }
    *have = grow;
    return TPC_MPC_OK;
}

// ---- shared between the C-ABI translation units (defined in tpc_mpc_api.cpp) --------------------
int check_common(tpc_mpc_context* h, const tpc_mpc_params* p, bool host_ok = false);
int check_compact_model(tpc_mpc_context* h, const tpc_mpc_params* p);
int stream_order_begin(tpc_mpc_context* h, hipStream_t s);
int stream_order_end(tpc_mpc_context* h, hipStream_t s);
int finish_flags(tpc_mpc_context* h, uint32_t* flags_out, hipStream_t s);
// stream_order_begin .. stream_order_end around everything an entry point enqueues.  The event behind the
// work is recorded on EVERY path out of the scope -- an error return in the middle leaves copies, memsets or
// kernels of this call on the stream, and the next solve (possibly on another stream) must still wait for them.
struct StreamOrderScope {
    tpc_mpc_context* h;
    hipStream_t s;
    bool armed = false;
    StreamOrderScope(tpc_mpc_context* hh, hipStream_t ss) : h(hh), s(ss) {}
    int begin() { const int rc = stream_order_begin(h, s); armed = rc == 0; return rc; }
    int end() { armed = false; return stream_order_end(h, s); }
    ~StreamOrderScope() { if (armed) (void)stream_order_end(h, s); }
};
// n instances of the compact form, arrays in DEVICE memory, launches only (no flag reset, no
// stream-order bookkeeping, no synchronisation): the core of every compact entry point.
int general_launch(tpc_mpc_context* h, const tpc_mpc_params* p, const tpc_mpc_general_io* io, hipStream_t s);
int check_general_device_io(tpc_mpc_context* h, const tpc_mpc_general_io* io);
int compact_launch(tpc_mpc_context* h, const tpc_mpc_params* p, int64_t n, const void* v, const void* dy,
                   const void* dphi, void* front, void* rear, int32_t* iters, hipStream_t s);
// AUTO's presolve around a compact solve whose tolerance pass the caller launches itself (tpc_mpc_mixed.hip starts the
// longest bin's presolve before the first bin and finishes it behind the last): begin forks the handle's side stream and
// launches the bit-exact kernels there for the instances lambda predicts to end on the cap; compact_launch(..., ps) then
// runs the tolerance pass only; finish joins, merges the side results over the pass's and runs the flag-driven second pass
// for what the prediction missed.  ps->on false: nothing was started (not applicable) and finish does the plain second pass.
struct Presolve {
    bool on = false;
    bool deferred = false;      // compact_launch leaves merge + second pass to presolve_finish
    double lambda_from = 0.0;
    uint32_t limit = 0;         // the longest queue the presolve kernel takes (one round of its wavefronts)
    int group_waves = 0;        // what is left of the chip for the tolerance family's persistent grid meanwhile
    uint32_t* queue = nullptr;  // instances taken, and how many (device)
    uint32_t* queue_len = nullptr;
    void *side_front = nullptr, *side_rear = nullptr;
    int32_t* side_iters = nullptr;
    int32_t* select = nullptr;  // the tolerance pass's iteration counts (what the second pass selects on)
};
int presolve_begin(tpc_mpc_context* h, const tpc_mpc_params* p, int64_t n, const void* v, const void* dy, const void* dphi,
                   hipStream_t s, Presolve* ps);
int compact_launch_ps(tpc_mpc_context* h, const tpc_mpc_params* p, int64_t n, const void* v, const void* dy,
                      const void* dphi, void* front, void* rear, int32_t* iters, hipStream_t s, Presolve* ps);
int presolve_finish(tpc_mpc_context* h, const tpc_mpc_params* p, int64_t n, const void* v, const void* dy, const void* dphi,
                    void* front, void* rear, int32_t* iters, hipStream_t s, Presolve* ps);
bool group_applicable(const tpc_mpc_context* h, const tpc_mpc_params* p, int H);
// scratch of the LANE family for (H, dtype, n), without launching (grows the handle's workspace)
int reserve_lane_workspace(tpc_mpc_context* h, int H, int dtype, int64_t n);

// a handle's device-side state for `device` (what tpc_mpc_create makes); the caller has checked the device
int context_new(int device, int cu_count, tpc_mpc_context** out);
// tpc_mpc_one.hip
int one_shot_solve(tpc_mpc_context* h, const tpc_mpc_params* p, double v, double dy, double dphi, double* front,
                   double* rear);
void one_shot_destroy(tpc_mpc_context* h);
// tpc_mpc_host.cpp: one compact instance on the calling thread (0 = solved, -1 = not a request it takes)
int host_solve_one(const tpc_mpc_params* p, double v, double dy, double dphi, double* front, double* rear, int* iters,
                   unsigned* flags);
// tpc_mpc_comm.cpp
void comm_destroy(tpc_mpc_context* h);

}  // namespace tpc
