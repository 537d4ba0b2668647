// tpc_mpc_solve_one: the reference's real configuration is ONE horizon-4 solve per cycle()
// (reference: include/trajectory_point_follower.h:48, src/trajectory_point_follower.cpp:366-380),
// about 5 us of dlib on a CPU core.  A kernel launch plus a stream synchronisation alone cost several
// times that, so single solves are served by a RESIDENT kernel: one wavefront that stays on the GPU
// and takes requests through a mailbox in pinned host memory mapped into the device.
//
//   host   writes the request (solver knobs, model constants, v, dy, dphi) into three 64-byte request
//          lines, each ending in the request's sequence number, and presets the two output words of
//          the mailbox to a NaN no solve produces.  The request lines live in DEVICE memory where the
//          CPU can write it (large BAR: stores through a write-combining mapping, then a store fence),
//          so that the wave polls its own HBM and the request costs a posted write instead of a PCIe
//          read round trip (measured: 1.75 us per echo against 2.39 us); otherwise in the pinned block
//   wave   polls the three lines with ONE wave-wide uncached load per poll (24 lanes x 8 bytes); a
//          request is taken when all three lines carry the same new sequence number (a line is read
//          as it stood at one instant and the host writes a line's number last, so a line with the
//          new number carries the new payload); the fields go from the polled registers straight into
//          the WAVE solve (mpc_wave.h: one decision variable per lane), which stores front and rear
//          straight into the mailbox in pinned host memory
//   host   spins on the two output words
//
// so a solve costs a posted write (or one PCIe read round trip), the solve, and one posted write: no
// launch, no synchronisation call.
//
// The wave never outlives its use: it exits when told to (tpc_mpc_destroy), after `idle_us` without
// a request (default 20 ms: longer than the cycle of a 50 Hz control loop, short enough that a
// device-wide synchronisation elsewhere in the process -- which has to wait for every running
// kernel -- is not held up noticeably), and after a fixed number of polls whatever the clocks say.
// The host notices (`alive` word) and starts a new one with the next request.  Horizons without a
// specialised kernel, fp32 and TPC_MPC_ALGO_LANE go through an ordinary launch.
#include "tpc_mpc_context.h"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "mpc_wave.h"

namespace tpc {

namespace {

// ---- mailbox layout: 8-byte words of the handle's 512-byte pinned block (the launch path uses words 32..) -----------------------
// line 0 (host -> device): words 0..6 payload, word 7 seq
// line 1 (host -> device): words 8..14 payload, word 15 seq
// line 2 (host -> device): words 16..22 payload, word 23 seq
// line 3 (device -> host): word 24 front, word 25 rear, word 26 alive, word 27 info = (TPC_MPC_FLAG_* << 32) | iterations
enum : int {
    kW_HorizonQuit = 0,   // low 32 bits horizon, high 32 bits quit flag
    kW_Iters = 1,         // low 32 bits max_iter, high 32 bits smo_iters
    kW_Eps = 2, kW_Step = 3, kW_Wheelbase = 4, kW_Q0 = 5, kW_Q1 = 6, kW_Seq0 = 7,
    kW_R0 = 8, kW_R1 = 9, kW_Lo0 = 10, kW_Lo1 = 11, kW_Hi0 = 12, kW_Hi1 = 13, kW_V = 14, kW_Seq1 = 15,
    kW_Dy = 16, kW_Dphi = 17, kW_Seq2 = 23,
    kW_Front = 24, kW_Rear = 25, kW_Alive = 26, kW_Info = 27,
    kW_Timing = 28,       // 28..30: the -DTPC_ONE_TIMING diagnostic build's counters
    kReqWords = 24,
};
constexpr uint64_t kSentinel = 0x7ff8dead5eedc0deull;   // a NaN payload no solve produces
constexpr uint32_t kMaxPolls = 1u << 26;                // backstop: ~1 us per poll -> about a minute

TPC_DEV uint64_t sys_load(const uint64_t* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
TPC_DEV void sys_store(uint64_t* p, uint64_t x) {
    __hip_atomic_store(p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// One instantiation per horizon: a resident wave serves the horizon it was started for (a module
// uses one), and the host swaps it for another when a request names a different horizon.  (One kernel
// switching over all horizons carried the largest one's 256 registers plus 500 spilled SGPRs into
// every solve: slower than an ordinary launch at N = 20.)
template <int H>
__global__ __launch_bounds__(64) void one_shot_kernel(const uint64_t* req, uint64_t* mail, uint64_t start_seq, uint64_t idle_ticks) {
    __shared__ __attribute__((aligned(16))) double s_w[2 * H];
    __shared__ __attribute__((aligned(16))) double s_row1[wave_row_lds<double, 2, H, CompactModel<double>>()];   // (N = 40: a Hessian row, mpc_wave.h)
    const int lane = threadIdx.x;
    uint64_t seen = start_seq;
    uint64_t idle_since = wall_clock64();   // 100 MHz
    // The set-up a solve derives from the model alone (Hessian row, Q_diag, lambda, 1 / lambda, beta: ~300 of the ~450
    // instructions before the first iteration) stays in registers between requests and is reused while a request repeats
    // the previous one's v and parameters -- the words of the request lines below, compared bit for bit (eps too: the
    // stop-test screen looks at it).  A vehicle at constant speed, or the target sweep of one cycle, hits; a changed v
    // recomputes.  Up to N = 20 (beyond, the row does not fit beside the solve's registers).
    constexpr bool kKeep = H <= 20;
    constexpr uint64_t kModelWords = (1ull << kW_Eps) | (1ull << kW_Step) | (1ull << kW_Wheelbase) | (1ull << kW_Q0) |
                                     (1ull << kW_Q1) | (1ull << kW_R0) | (1ull << kW_R1) | (1ull << kW_Lo0) |
                                     (1ull << kW_Lo1) | (1ull << kW_Hi0) | (1ull << kW_Hi1) | (1ull << kW_V);
    WaveKeep<double, kKeep ? H : 1> keep;
    uint64_t kept_word = 0;
    bool kept = false;
    for (uint32_t polls = 0; polls < kMaxPolls; ++polls) {
        // one wave-wide uncached read of the three request lines
        const uint64_t word = lane < kReqWords ? sys_load(req + lane) : 0ull;
        // a field of the request = one lane of `word`, read with v_readlane (a register move to the scalar
        // side; the shuffle intrinsic goes through the LDS crossbar, ~20x the latency)
        auto field = [&](int w) {
            const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)word, w);
            const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(word >> 32), w);
            return ((uint64_t)hi << 32) | lo;
        };
        const uint64_t q0 = field(kW_Seq0), q1 = field(kW_Seq1), q2 = field(kW_Seq2);
        if (q0 == seen || q0 != q1 || q0 != q2) {
            if (wall_clock64() - idle_since > idle_ticks) break;
            continue;
        }
        seen = q0;
        auto real = [&](int w) { return __longlong_as_double((long long)field(w)); };
        const uint64_t hq = field(kW_HorizonQuit);
        if ((uint32_t)(hq >> 32) != 0u) break;   // told to quit
        if ((int)(uint32_t)hq == H) {
            OneArgs g;
            g.v = real(kW_V); g.dy = real(kW_Dy); g.dphi = real(kW_Dphi);
            g.step = real(kW_Step); g.wheelbase = real(kW_Wheelbase);
            g.q[0] = real(kW_Q0); g.q[1] = real(kW_Q1); g.r[0] = real(kW_R0); g.r[1] = real(kW_R1);
            g.lo[0] = real(kW_Lo0); g.lo[1] = real(kW_Lo1); g.hi[0] = real(kW_Hi0); g.hi[1] = real(kW_Hi1);
            g.out = mail + kW_Front;   // two 8-byte stores, each atomic for the host: with the info word they are the completion signal
            g.info = mail + kW_Info;
            Knobs kn;
            kn.eps = real(kW_Eps);
            const uint64_t its = field(kW_Iters);
            kn.max_iter = (uint32_t)its;
            kn.smo_iters = (uint32_t)(its >> 32);
#ifdef TPC_ONE_TIMING   // diagnostic build only (scripts/latency.sh timing): where a solve's time goes, in shader clocks
            const uint64_t t_seen = clock64();
            uint64_t t_loops = 0;
            auto stamp = [&]() { t_loops = clock64(); };
#else
            NoHook stamp;
#endif
            if constexpr (kKeep) {
                keep.hit = kept && __ballot(((kModelWords >> lane) & 1ull) != 0ull && word != kept_word) == 0ull;
                wave_solve_any<double, 2, H, CompactModel<double>, OneArgs, decltype(stamp), WaveKeep<double, H>>(g, kn, 0, s_w, s_row1, stamp, &keep);
                kept_word = word;
                kept = true;
            } else {
                wave_solve_any<double, 2, H, CompactModel<double>, OneArgs, decltype(stamp)>(g, kn, 0, s_w, s_row1, stamp);
            }
#ifdef TPC_ONE_TIMING
            const uint64_t t_done = clock64();
            if (lane == 0) {
                sys_store(mail + kW_Timing, sys_load(mail + kW_Timing) + (t_loops - t_seen));
                sys_store(mail + kW_Timing + 1, sys_load(mail + kW_Timing + 1) + (t_done - t_loops));
                sys_store(mail + kW_Timing + 2, sys_load(mail + kW_Timing + 2) + 1);
            }
#endif
        } else if (lane < 2) {
            sys_store(mail + kW_Front + lane, 0x7ff8000000000badull);   // the host never asks this
            if (lane == 0) sys_store(mail + kW_Info, 0ull);
        }
        idle_since = wall_clock64();
        polls = 0;
    }
    if (lane == 0) sys_store(mail + kW_Alive, 0ull);
}

}  // namespace

struct OneShot {
    hipStream_t stream = nullptr;          // the resident wave's
    hipStream_t launch_stream = nullptr;   // ordinary single-solve launches (horizons / dtypes without a resident kernel)
    uint64_t seq = 0;
    int horizon = 0;         // of the resident wave (0: none was started yet)
    uint64_t idle_us = 20000;
    bool disabled = false;   // set after a resident kernel failed to answer: ordinary launches from then on
    uint64_t* req_dev = nullptr;   // the request lines in device memory the CPU can write (large BAR), or null
};

namespace {

inline volatile uint64_t* mailbox(tpc_mpc_context* h) { return (volatile uint64_t*)h->pin_host; }
// where the host writes a request: device memory through the BAR when there is any, else the mailbox itself
inline volatile uint64_t* request_lines(tpc_mpc_context* h) {
    return h->one && h->one->req_dev ? (volatile uint64_t*)h->one->req_dev : mailbox(h);
}
inline void publish_request() {
#if defined(__x86_64__)
    __builtin_ia32_sfence();               // drains the write-combining buffers of a BAR mapping
#else
    __atomic_thread_fence(__ATOMIC_SEQ_CST);
#endif
}

int start_kernel(tpc_mpc_context* h, OneShot* o, int horizon, uint64_t start_seq) {
    HIP_TRY(h, hipSetDevice(h->device));
    volatile uint64_t* m = mailbox(h);
    m[kW_Alive] = 1;
    __atomic_thread_fence(__ATOMIC_SEQ_CST);
    uint64_t* dev = (uint64_t*)h->pin_dev;
    const uint64_t* req = o->req_dev ? o->req_dev : dev;
    const uint64_t ticks = o->idle_us * 100ull;
    switch (horizon) {
#define X(hh) case hh: hipLaunchKernelGGL(one_shot_kernel<hh>, dim3(1), dim3(kWave), 0, o->stream, req, dev, start_seq, ticks); break;
        X(4) X(5) X(10) X(20) X(30) X(40)
#undef X
        default: m[kW_Alive] = 0; return fail(h, TPC_MPC_ERR_BAD_HORIZON, "no resident kernel for horizon %d", horizon);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { m[kW_Alive] = 0; return hip_fail(h, e, "resident kernel launch"); }
    o->horizon = horizon;
    return TPC_MPC_OK;
}

// Ask the running wave to leave (a request whose quit flag is set) and wait for it.
int stop_kernel(tpc_mpc_context* h, OneShot* o) {
    volatile uint64_t* m = mailbox(h);
    if (o->stream && m[kW_Alive]) {
        HIP_TRY(h, hipSetDevice(h->device));
        volatile uint64_t* rq = request_lines(h);
        const uint64_t seq = ++o->seq;
        rq[kW_HorizonQuit] = 1ull << 32;
        // the flag before the numbers, for real: a release fence orders nothing among write-combined BAR
        // stores, the store fence drains them (as one_shot_solve does)
        __atomic_thread_fence(__ATOMIC_SEQ_CST);
        publish_request();
        rq[kW_Seq0] = seq; rq[kW_Seq1] = seq; rq[kW_Seq2] = seq;
        publish_request();
        HIP_TRY(h, hipStreamSynchronize(o->stream));   // bounded: quit flag, idle timeout, poll cap
    }
    return TPC_MPC_OK;
}

// The pre-resident path: one WAVE/LANE launch through the handle's mapped block, polled the same way.
int launch_path(tpc_mpc_context* h, const tpc_mpc_params* p, double v, double dy, double dphi, double* front, double* rear) {
    HIP_TRY(h, hipSetDevice(h->device));
    tpc_mpc_params q = *p;
    // one instance: one wavefront, where a WAVE kernel exists; AUTO sorts out the other horizons itself
    const size_t es = esize(q.dtype);
    char* hp = (char*)h->pin_host + 32 * 8;   // words 32..: apart from the resident kernel's lines
    char* dp = (char*)h->pin_dev + 32 * 8;
    // wait for no resident solve here: they are synchronous, none is in flight
    if (q.dtype == TPC_MPC_F64) { double* x = (double*)hp; x[0] = v; x[1] = dy; x[2] = dphi; }
    else { float* x = (float*)hp; x[0] = (float)v; x[1] = (float)dy; x[2] = (float)dphi; }
    const uint64_t sentinel64 = kSentinel;
    const uint32_t sentinel32 = 0x7fc5eed1u;
    if (q.dtype == TPC_MPC_F64) { std::memcpy(hp + 3 * es, &sentinel64, 8); std::memcpy(hp + 4 * es, &sentinel64, 8); }
    else { std::memcpy(hp + 3 * es, &sentinel32, 4); std::memcpy(hp + 4 * es, &sentinel32, 4); }
    // on the handle's own non-blocking stream, not the NULL stream (which would serialise with every
    // blocking stream of the process); ordered against the handle's other solves like any solve
    // (a stream of its own: the resident wave occupies the other one for as long as it lives)
    OneShot* o = h->one;
    if (!o->launch_stream) HIP_TRY(h, hipStreamCreateWithFlags(&o->launch_stream, hipStreamNonBlocking));
    // the iteration count lands in the pinned block like the outputs; the flag word of the handle is copied
    // there behind the kernels (words 40, 41 of the block: bytes 64.. of this path's half)
    volatile uint32_t* extra = (volatile uint32_t*)(hp + 64);
    const uint32_t sentinel_flags = 0xffffffffu;
    extra[0] = 0; extra[2] = sentinel_flags;
    __atomic_thread_fence(__ATOMIC_SEQ_CST);
    StreamOrderScope order(h, o->launch_stream);
    int rc = order.begin();
    if (rc) return rc;
    HIP_TRY(h, hipMemsetAsync(h->ws_words + 1, 0, sizeof(uint32_t), o->launch_stream));
    rc = compact_launch(h, &q, 1, dp, dp + es, dp + 2 * es, dp + 3 * es, dp + 4 * es, (int32_t*)(dp + 64), o->launch_stream);
    if (rc) return rc;
    HIP_TRY(h, hipMemcpyAsync((void*)(hp + 72), h->ws_words + 1, sizeof(uint32_t), hipMemcpyDeviceToHost, o->launch_stream));
    rc = order.end();
    if (rc) return rc;
    bool done = false;
    const auto t0 = std::chrono::steady_clock::now();
    for (int it = 0; !done; ++it) {
        if (q.dtype == TPC_MPC_F64) {
            const volatile uint64_t* o = (const volatile uint64_t*)(hp + 3 * es);
            done = o[0] != sentinel64 && o[1] != sentinel64;
        } else {
            const volatile uint32_t* o = (const volatile uint32_t*)(hp + 3 * es);
            done = o[0] != sentinel32 && o[1] != sentinel32;
        }
        done = done && extra[2] != sentinel_flags;
        if (!done && (it & 255) == 255 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) break;
    }
    if (!done) HIP_TRY(h, hipStreamSynchronize(o->launch_stream));
    h->one_flags = extra[2];
    h->one_iters = (int32_t)extra[0];
    h->one_valid = true;
    if (q.dtype == TPC_MPC_F64) { *front = ((double*)hp)[3]; *rear = ((double*)hp)[4]; }
    else { *front = ((float*)hp)[3]; *rear = ((float*)hp)[4]; }
    return TPC_MPC_OK;
}

}  // namespace

void one_shot_destroy(tpc_mpc_context* h) {
    if (!h || !h->one) return;
    OneShot* o = h->one;
#ifdef TPC_ONE_TIMING
    {
        volatile uint64_t* m = mailbox(h);
        if (m[kW_Timing + 2]) fprintf(stderr, "[one timing] %llu solves: request seen -> loops %.0f clocks, loops -> result stored %.0f clocks\n",
                           (unsigned long long)m[kW_Timing + 2], (double)m[kW_Timing] / m[kW_Timing + 2], (double)m[kW_Timing + 1] / m[kW_Timing + 2]);
    }
#endif
    if (o->stream) {
        (void)stop_kernel(h, o);
        (void)hipStreamSynchronize(o->stream);
        (void)hipStreamDestroy(o->stream);
    }
    if (o->launch_stream) {
        (void)hipStreamSynchronize(o->launch_stream);
        (void)hipStreamDestroy(o->launch_stream);
    }
    if (o->req_dev) (void)hipFree(o->req_dev);
    delete o;
    h->one = nullptr;
}

int one_shot_configure(tpc_mpc_context* h, int64_t idle_us) {
    HIP_TRY(h, hipSetDevice(h->device));
    if (!h->one) {
        h->one = new (std::nothrow) OneShot;
        if (!h->one) return fail(h, TPC_MPC_ERR_ALLOC, "out of host memory");
        // Request lines in device memory, if the CPU can write it: every byte of VRAM sits behind the PCIe BAR
        // ("large BAR"), and a fine-grained allocation is mapped for the host as well.  TPC_MPC_OPT_MAILBOX_HOST
        // keeps them in the pinned block (the fallback on a part without a large BAR).
        int large_bar = 0;
        if (!h->opt_mailbox_host &&
            hipDeviceGetAttribute(&large_bar, hipDeviceAttributeIsLargeBar, h->device) == hipSuccess && large_bar) {
            void* p = nullptr;
            if (hipExtMallocWithFlags(&p, 256, hipDeviceMallocFinegrained) == hipSuccess) {
                volatile uint64_t* z = (volatile uint64_t*)p;   // cleared from here, through the mapping the requests will use
                for (int w = 0; w < 32; ++w) z[w] = 0;
                publish_request();
                h->one->req_dev = (uint64_t*)p;
            }
            (void)hipGetLastError();   // a refusal here only means the pinned block is used
        }
    }
    OneShot* o = h->one;
    h->one_idle_us = idle_us;   // kept on the handle: a mailbox restart (TPC_MPC_OPT_MAILBOX_HOST) must not forget it
    if (idle_us <= 0) {   // resident mode off: stop a running wave, keep the launch path
        int rc = stop_kernel(h, o);
        if (rc) return rc;
        o->disabled = true;
        return TPC_MPC_OK;
    }
    o->disabled = false;
    o->idle_us = (uint64_t)idle_us;
    return TPC_MPC_OK;
}

int one_shot_solve(tpc_mpc_context* h, const tpc_mpc_params* p, double v, double dy, double dphi, double* front, double* rear) {
    if (!h->one) {
        int rc = one_shot_configure(h, h->one_idle_us);   // (what tpc_mpc_set_resident last asked for; 20 ms by default)
        if (rc) return rc;
    }
    OneShot* o = h->one;
    const int H = p->horizon;   // resident kernels exist for the specialised horizons the WAVE layout can take
    const bool resident_ok = !o->disabled && p->dtype == TPC_MPC_F64 && p->algo != TPC_MPC_ALGO_LANE &&
                             (H == 4 || H == 5 || H == 10 || H == 20 || H == 30 || H == 40);
    if (!resident_ok) return launch_path(h, p, v, dy, dphi, front, rear);
    if (!o->stream) {
        HIP_TRY(h, hipSetDevice(h->device));
        HIP_TRY(h, hipStreamCreateWithFlags(&o->stream, hipStreamNonBlocking));
    }
    volatile uint64_t* m = mailbox(h);
    if (o->horizon != p->horizon && o->horizon != 0) {   // the resident wave serves another horizon: swap it
        int rc = stop_kernel(h, o);
        if (rc) return rc;
    }
    volatile uint64_t* rq = request_lines(h);   // write-only for the host when it is device memory (reads over the BAR are slow)
    auto put = [&](int w, double x) { uint64_t b; std::memcpy(&b, &x, 8); rq[w] = b; };
    const uint64_t prev = o->seq, seq = ++o->seq;
    m[kW_Front] = kSentinel; m[kW_Rear] = kSentinel; m[kW_Info] = kSentinel;
    rq[kW_HorizonQuit] = (uint64_t)(uint32_t)p->horizon;
    rq[kW_Iters] = (uint64_t)(uint32_t)p->max_iter | ((uint64_t)(uint32_t)p->smo_iters << 32);
    put(kW_Eps, p->eps); put(kW_Step, p->step_size); put(kW_Wheelbase, p->wheelbase);
    put(kW_Q0, p->weight_y); put(kW_Q1, p->weight_phi);
    put(kW_R0, p->weight_steering_front); put(kW_R1, p->weight_steering_rear);
    put(kW_Lo0, p->lower[0]); put(kW_Lo1, p->lower[1]); put(kW_Hi0, p->upper[0]); put(kW_Hi1, p->upper[1]);
    put(kW_V, v); put(kW_Dy, dy); put(kW_Dphi, dphi);
    // the sentinels and the payload before the numbers (x86 keeps the order of stores of one memory type; the
    // fence orders the pinned block's sentinels against the write-combined request and stops the compiler)
    __atomic_thread_fence(__ATOMIC_SEQ_CST);
    rq[kW_Seq0] = seq; rq[kW_Seq1] = seq; rq[kW_Seq2] = seq;
    publish_request();
    if (!m[kW_Alive]) {
        int rc = start_kernel(h, o, p->horizon, prev);   // it finds the request already waiting
        if (rc) return rc;
    }
    const auto t0 = std::chrono::steady_clock::now();
    int restarts = 0;
    auto answered = [&]() { return m[kW_Front] != kSentinel && m[kW_Rear] != kSentinel && m[kW_Info] != kSentinel; };
    for (uint32_t it = 1;; ++it) {
        if (answered()) break;
        if ((it & 1023u) != 0) continue;
        if (!m[kW_Alive]) {
            // the wave left (idle timeout) just as the request was posted: start another one for it
            if (answered()) break;
            if (++restarts > 3) { o->disabled = true; return launch_path(h, p, v, dy, dphi, front, rear); }
            int rc = start_kernel(h, o, p->horizon, prev);
            if (rc) return rc;
        }
        if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) {
            // Never spin for ever.  The wave may still be alive and working on this very request: it is told to
            // leave and WAITED FOR (bounded: quit flag, idle timeout, poll cap) before anything else happens, so
            // that no late answer of it can ever be taken for the answer to a later request; the handle then
            // solves through ordinary launches, starting with this request.
            o->disabled = true;
            int rc = stop_kernel(h, o);
            if (rc) return rc;
            if (answered()) break;   // it did answer in the end
            return launch_path(h, p, v, dy, dphi, front, rear);
        }
    }
    uint64_t fb = m[kW_Front], rb = m[kW_Rear];
    std::memcpy(front, &fb, 8);
    std::memcpy(rear, &rb, 8);
    const uint64_t info = m[kW_Info];
    h->one_flags = (uint32_t)(info >> 32);
    h->one_iters = (int32_t)(uint32_t)info;
    h->one_valid = true;
    // AUTO's guarantee (include/tpc_mpc.h): a solve the resident wavefront (WAVE arithmetic) left on the iteration cap is
    // solved once more in dlib's own operation order -- one LANE launch; the flag came with the answer, so it costs
    // nothing to know
    if ((h->one_flags & TPC_MPC_FLAG_MAX_ITER) && p->algo == TPC_MPC_ALGO_AUTO && p->max_iter > 0 &&
        (p->options & TPC_MPC_PARAM_FAST_CAPPED) == 0) {
        tpc_mpc_params q = *p;
        q.algo = TPC_MPC_ALGO_LANE;
        return launch_path(h, &q, v, dy, dphi, front, rear);
    }
    return TPC_MPC_OK;
}

}  // namespace tpc

using namespace tpc;

extern "C" int tpc_mpc_set_resident(tpc_mpc_handle h, int64_t idle_timeout_us) {
    return guarded(h, [&]() -> int {
        if (!h) return fail(nullptr, TPC_MPC_ERR_BAD_ARG, "null handle");
        if (h->host_only) return fail(h, TPC_MPC_ERR_NO_DEVICE, "host-only handle");
        if (idle_timeout_us > 10 * 1000 * 1000) return fail(h, TPC_MPC_ERR_BAD_ARG, "idle timeout above 10 s");
        HIP_TRY(h, hipSetDevice(h->device));
        return one_shot_configure(h, idle_timeout_us);
    });
}
