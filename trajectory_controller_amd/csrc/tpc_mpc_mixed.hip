// tpc_mpc_solve_batch_compact_mixed: one batch whose instances carry their own horizon (BASELINE.json
// config 5: N in {5, 10, 20, 40} mixed in one batch).  The solver kernels are specialised per horizon,
// so the batch is binned by horizon ON THE DEVICE, each bin is solved by one launch sequence of its
// horizon's kernels on bin-contiguous copies of the inputs, and the outputs are scattered back to the
// caller's order.  Integer/byte work around the solves, HBM-bound and tiny next to them:
//
//   mixed_bin_kernel      bin index of every instance + its slot inside the bin.  One ballot per bin
//                         and ONE atomic per (wavefront, bin): a quarter of a million single atomics
//                         on six counters would serialise (the same effect the queue histogram of
//                         mpc_lane.h avoids).  [n x 4 B read, n x 4 B written]
//   (host)                reads the six bin sizes -- the one synchronisation of the call -- and turns
//                         them into bin offsets
//   mixed_gather_kernel   inputs -> bin-contiguous arrays, perm[] remembers where each came from
//   (per bin)             the ordinary compact solve of that horizon (compact_launch)
//   mixed_scatter_kernel  outputs back into the caller's order
#include "tpc_mpc_context.h"

namespace tpc {

namespace {

constexpr int kBinsMax = 8;   // six supported horizons + "unsupported" + padding

struct MixedBins {
    int nb;               // supported horizons
    int horizon[kBinsMax];
};

__device__ __forceinline__ int bin_of(const MixedBins& b, int h) {
    int r = b.nb;   // unsupported
#pragma unroll
    for (int i = 0; i < kBinsMax - 1; ++i)
        if (i < b.nb && b.horizon[i] == h) r = i;
    return r;
}

// counts[0..nb]: instances per bin (nb = unsupported horizon); slot[k]: position inside the bin
__global__ __launch_bounds__(256) void mixed_bin_kernel(const int32_t* __restrict__ horizons, int64_t n, MixedBins bins,
                                                        uint32_t* __restrict__ counts, uint32_t* __restrict__ slot) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = k < n;
    const int b = live ? bin_of(bins, horizons[k]) : -1;
    const int lane = threadIdx.x & (kWave - 1);
    uint32_t mine = 0;
    for (int i = 0; i <= bins.nb; ++i) {
        const unsigned long long m = __ballot(b == i);
        if (m == 0ull) continue;
        const int leader = __ffsll((long long)m) - 1;
        uint32_t base = 0;
        if (lane == leader) base = atomicAdd(&counts[i], (uint32_t)__popcll(m));
        base = (uint32_t)__shfl((int)base, leader);
        if (b == i) mine = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    }
    if (live) slot[k] = mine;
}

template <typename T>
__global__ __launch_bounds__(256) void mixed_gather_kernel(const int32_t* __restrict__ horizons, int64_t n, MixedBins bins,
                                                           const uint32_t* __restrict__ slot, const uint32_t* __restrict__ offsets,
                                                           const T* __restrict__ v, const T* __restrict__ dy,
                                                           const T* __restrict__ dphi, T* __restrict__ gv,
                                                           T* __restrict__ gdy, T* __restrict__ gdphi,
                                                           uint32_t* __restrict__ perm) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const int b = bin_of(bins, horizons[k]);
    if (b >= bins.nb) return;   // cannot happen: the call was refused on the host
    const uint32_t pos = offsets[b] + slot[k];
    perm[pos] = (uint32_t)k;
    gv[pos] = v[k];
    gdy[pos] = dy[k];
    gdphi[pos] = dphi[k];
}

template <typename T>
__global__ __launch_bounds__(256) void mixed_scatter_kernel(int64_t n, const uint32_t* __restrict__ perm,
                                                            const T* __restrict__ gfront, const T* __restrict__ grear,
                                                            const int32_t* __restrict__ giters, T* __restrict__ front,
                                                            T* __restrict__ rear, int32_t* __restrict__ iters) {
    const int64_t pos = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (pos >= n) return;
    const uint32_t k = perm[pos];
    front[k] = gfront[pos];
    rear[k] = grear[pos];
    if (iters) iters[k] = giters[pos];
}

__global__ void mixed_or_flags_kernel(uint32_t* __restrict__ dst, const uint32_t* __restrict__ src) {
    const uint32_t f = *src;
    if (f) atomicOr(dst, f);
}

template <typename T>
int run_mixed(tpc_mpc_context* h, const tpc_mpc_params* p, int64_t n, const int32_t* d_hz, const T* d_v, const T* d_dy,
              const T* d_dphi, T* d_front, T* d_rear, int32_t* d_iters, char* scratch, hipStream_t s) {
    // scratch: counts[8] | offsets[8] | slot[n] | perm[n] | gv | gdy | gdphi | gfront | grear | giters
    const int64_t col = pad256(n * (int64_t)sizeof(T)), icol = pad256(n * 4);
    uint32_t* counts = (uint32_t*)scratch;
    uint32_t* offsets = counts + kBinsMax;
    uint32_t* slot = (uint32_t*)(scratch + 256);
    uint32_t* perm = (uint32_t*)(scratch + 256 + icol);
    char* g = scratch + 256 + 2 * icol;
    T *gv = (T*)g, *gdy = (T*)(g + col), *gdphi = (T*)(g + 2 * col), *gfront = (T*)(g + 3 * col), *grear = (T*)(g + 4 * col);
    int32_t* giters = (int32_t*)(g + 5 * col);

    // a pending queue-order hint was given for some batch of n instances, not for a bin that happens to have
    // that size: forgotten here
    h->hint = nullptr;
    h->hint_n = 0;
    MixedBins bins;
    int hz[kBinsMax];
    bins.nb = tpc_mpc_supported_horizons(hz, kBinsMax - 1);
    for (int i = 0; i < kBinsMax; ++i) bins.horizon[i] = i < bins.nb ? hz[i] : -1;

    const unsigned grid = (unsigned)((n + 255) / 256);
    HIP_TRY(h, hipMemsetAsync(counts, 0, 2 * kBinsMax * sizeof(uint32_t), s));
    hipLaunchKernelGGL(mixed_bin_kernel, dim3(grid), dim3(256), 0, s, d_hz, n, bins, counts, slot);
    HIP_TRY(h, hipGetLastError());
    uint32_t hc[kBinsMax] = {0};
    HIP_TRY(h, hipMemcpyAsync(hc, counts, sizeof(hc), hipMemcpyDeviceToHost, s));
    HIP_TRY(h, hipStreamSynchronize(s));   // the one synchronisation: bin sizes decide the launches
    if (hc[bins.nb] != 0)
        return fail(h, TPC_MPC_ERR_BAD_HORIZON, "%u instances carry an unsupported horizon (supported: 4 5 10 20 30 40)",
                    hc[bins.nb]);
    uint32_t ho[kBinsMax] = {0};
    for (int i = 1; i < kBinsMax; ++i) ho[i] = ho[i - 1] + hc[i - 1];
    HIP_TRY(h, hipMemcpyAsync(offsets, ho, sizeof(ho), hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL((mixed_gather_kernel<T>), dim3(grid), dim3(256), 0, s, d_hz, n, bins, (const uint32_t*)slot,
                       (const uint32_t*)offsets, d_v, d_dy, d_dphi, gv, gdy, gdphi, perm);
    HIP_TRY(h, hipGetLastError());
    // (every bin's scratch belongs to its own child handle and is grown by that bin's launch, before that bin's
    // kernels; growing frees and allocates device memory, which synchronises the device: only ever on a handle's
    // first batches)
    // The bins are independent batches, and a LANE pass over a bin smaller than the chip lasts as long as its
    // slowest instance on a fraction of the SIMDs (N = 40: 29 ms on a quarter of them at 16 384 instances): they
    // run CONCURRENTLY, each on a child handle (scratch, ticket and flag word of its own) and a stream of its own,
    // forked from the caller's stream here and joined back into it below; longest horizon first.
    // The bins are independent batches.  How they share the chip:
    //   * under AUTO (or GROUP) the bins of N = 10, 20, 30, 40 are GROUP's: persistent grids of one wavefront per SIMD
    //     whose time is the time of their slowest instance at a LONE wavefront's pace.  Such bins must not overlap: a
    //     second wavefront on a SIMD fills every gap the first one's dependent instructions leave and delays each of its
    //     instructions by about two cycles, and one delayed wavefront delays the kernel (at N = 40 every wavefront holds
    //     a 10 000-iteration instance).  Measured on config 5 (4 x 16 384 instances): all bins at once 9.3-9.6 ms -- the
    //     N = 40 bin 7.7 ms instead of the 4.8 it takes alone, whatever s_setprio says; the bins on disjoint SHARES of
    //     the wavefront slots (898 / 110 / 16, a quarter of a CU's LDS claimed per wavefront so that no CU takes five):
    //     8.5-8.8 ms, because the hardware still stacks wavefronts of different kernels on one SIMD of a CU while
    //     another idles (the same 898 wavefronts alone: 4.6 ms).  So they run ONE AFTER ANOTHER on the caller's stream,
    //     each with the whole chip, shortest horizon first;
    //   * otherwise (a family demanded explicitly) the bins run concurrently on child handles, as LANE passes over bins
    //     smaller than the chip want (each lasts as long as its slowest instance on a fraction of the SIMDs).
    int live = 0;
    bool any_group = false;
    for (int i = 0; i < bins.nb; ++i) {
        if (!hc[i]) continue;
        ++live;
        any_group = any_group || group_applicable(h, p, bins.horizon[i]);
    }
    if (!h->fork_ev) HIP_TRY(h, hipEventCreateWithFlags(&h->fork_ev, hipEventDisableTiming));
    if (any_group || live == 1) {
        // AUTO's presolve of the LONGEST horizon's bin -- the one whose cap-bound instances cost a 10 000-iteration chain --
        // starts here, on the handle's side stream and on SIMDs of its own, before the first bin, and is finished behind the
        // last; every persistent grid launched meanwhile keeps to the other SIMDs (tpc_mpc_api.cpp: presolve_begin)
        int last = -1;
        for (int i = 0; i < bins.nb; ++i)
            if (hc[i]) last = i;
        Presolve ps;
        tpc_mpc_params ql = *p;
        ql.horizon = bins.horizon[last];
        const int64_t ol = ho[last];
        int rc = presolve_begin(h, &ql, hc[last], gv + ol, gdy + ol, gdphi + ol, s, &ps);
        ps.deferred = true;
        auto bail = [&](int code) {   // the side stream got work: the caller's stream must not run ahead of it
            if (ps.on) { (void)hipEventRecord(h->pre_done, h->pre_stream); (void)hipStreamWaitEvent(s, h->pre_done, 0); h->pre_busy = false; }
            return code;
        };
        if (rc) return bail(rc);
        for (int i = 0; i < bins.nb; ++i) {   // ascending horizon
            if (!hc[i]) continue;
            tpc_mpc_params q = *p;
            q.horizon = bins.horizon[i];
            const int64_t o = ho[i];
            if (i == last) rc = compact_launch_ps(h, &q, hc[i], gv + o, gdy + o, gdphi + o, gfront + o, grear + o, d_iters ? giters + o : nullptr, s, &ps);
            else rc = compact_launch(h, &q, hc[i], gv + o, gdy + o, gdphi + o, gfront + o, grear + o, d_iters ? giters + o : nullptr, s);
            if (rc) return bail(rc);
        }
        rc = presolve_finish(h, &ql, hc[last], gv + ol, gdy + ol, gdphi + ol, gfront + ol, grear + ol, d_iters ? giters + ol : nullptr, s, &ps);
        if (rc) return bail(rc);
        if (live > 1) {   // several solves went through this handle: its kernel times / lane statistics describe the last bin only
            h->ev_valid = false;
            h->last_algo = kAlgoMixed;
        }
    } else {
        h->ev_valid = false;   // the solves run on child handles
        h->last_algo = kAlgoMixed;
        HIP_TRY(h, hipEventRecord(h->fork_ev, s));
        // Whatever goes wrong below, the caller's stream must wait for every child stream that already got work, or the
        // next call's gather (or a growing scratch buffer) could overtake a child still running on the old contents.
        bool forked[kBinsMax] = {false};
        auto join = [&]() {
            for (int i = 0; i < bins.nb; ++i)
                if (forked[i] && hipEventRecord(h->kid_done[i], h->kid_stream[i]) == hipSuccess)
                    (void)hipStreamWaitEvent(s, h->kid_done[i], 0);
        };
        for (int i = bins.nb - 1; i >= 0; --i) {   // longest horizon first
            if (!hc[i]) continue;
            tpc_mpc_params q = *p;
            q.horizon = bins.horizon[i];
            const int64_t o = ho[i];
            if (!h->kids[i]) {   // published only when complete: handle, stream and event
                tpc_mpc_context* kid = nullptr;
                hipStream_t ks = nullptr;
                hipEvent_t ke = nullptr;
                int rc = context_new(h->device, h->cu_count, &kid);
                if (rc) { join(); return fail(h, rc, "mixed batch: child handle: %s", g_create_error); }
                hipError_t e = hipStreamCreateWithFlags(&ks, hipStreamNonBlocking);
                if (e == hipSuccess) e = hipEventCreateWithFlags(&ke, hipEventDisableTiming);
                if (e != hipSuccess) {
                    if (ks) (void)hipStreamDestroy(ks);
                    (void)tpc_mpc_destroy(kid);
                    join();
                    return hip_fail(h, e, "mixed batch: child stream / event");
                }
                h->kids[i] = kid; h->kid_stream[i] = ks; h->kid_done[i] = ke;
            }
            tpc_mpc_context* kid = h->kids[i];
            kid->opt_wave_group = h->opt_wave_group;
            kid->opt_group_lanes = h->opt_group_lanes;
            hipStream_t ks = h->kid_stream[i];
            hipError_t e = hipStreamWaitEvent(ks, h->fork_ev, 0);
            if (e != hipSuccess) { join(); return hip_fail(h, e, "mixed batch: fork"); }
            forked[i] = true;
            e = hipMemsetAsync(kid->ws_words + 1, 0, sizeof(uint32_t), ks);
            if (e != hipSuccess) { join(); return hip_fail(h, e, "mixed batch: flag word"); }
            int rc = compact_launch(kid, &q, hc[i], gv + o, gdy + o, gdphi + o, gfront + o, grear + o,
                                    d_iters ? giters + o : nullptr, ks);
            if (rc) { join(); return fail(h, rc, "mixed batch, horizon %d: %s", q.horizon, kid->err); }
            hipLaunchKernelGGL(mixed_or_flags_kernel, dim3(1), dim3(1), 0, ks, h->ws_words + 1, (const uint32_t*)(kid->ws_words + 1));
            e = hipGetLastError();
            if (e != hipSuccess) { join(); return hip_fail(h, e, "mixed batch: flags"); }
        }
        {   // join: the caller's stream continues behind every child
            hipError_t e = hipSuccess;
            for (int i = 0; i < bins.nb && e == hipSuccess; ++i) {
                if (!forked[i]) continue;
                e = hipEventRecord(h->kid_done[i], h->kid_stream[i]);
                if (e == hipSuccess) e = hipStreamWaitEvent(s, h->kid_done[i], 0);
            }
            if (e != hipSuccess) { join(); return hip_fail(h, e, "mixed batch: join"); }
        }
    }
    hipLaunchKernelGGL((mixed_scatter_kernel<T>), dim3(grid), dim3(256), 0, s, n, (const uint32_t*)perm, (const T*)gfront,
                       (const T*)grear, (const int32_t*)giters, d_front, d_rear, d_iters);
    HIP_TRY(h, hipGetLastError());
    return TPC_MPC_OK;
}

}  // namespace
}  // namespace tpc

using namespace tpc;

extern "C" int tpc_mpc_solve_batch_compact_mixed(tpc_mpc_handle h, const tpc_mpc_params* p, int64_t n,
                                                 const int32_t* horizons, const void* v, const void* delta_y,
                                                 const void* delta_phi, void* steering_front, void* steering_rear,
                                                 int32_t* iters, uint32_t* flags_out, int mem, void* stream) {
    return guarded(h, [&]() -> int {
        if (!h) return fail(nullptr, TPC_MPC_ERR_BAD_ARG, "null handle");
        if (!p) return fail(h, TPC_MPC_ERR_BAD_ARG, "null params");
        tpc_mpc_params q = *p;
        int hz[8];
        (void)tpc_mpc_supported_horizons(hz, 8);
        q.horizon = hz[0];   // p->horizon is ignored: every instance names its own
        int rc = check_common(h, &q);
        if (rc) return rc;
        rc = check_compact_model(h, &q);
        if (rc) return rc;
        if (n < 0 || n > 0x7fffffffll) return fail(h, TPC_MPC_ERR_BAD_ARG, "need 0 <= n < 2^31");
        if (mem != TPC_MPC_HOST && mem != TPC_MPC_DEVICE) return fail(h, TPC_MPC_ERR_BAD_ARG, "bad memory kind");
        if (n == 0) { if (flags_out) *flags_out = 0; return TPC_MPC_OK; }
        if (!horizons || !v || !delta_y || !delta_phi || !steering_front || !steering_rear)
            return fail(h, TPC_MPC_ERR_BAD_ARG, "null batch pointer");
        HIP_TRY(h, hipSetDevice(h->device));
        hipStream_t s = (hipStream_t)stream;
        const int64_t es = (int64_t)esize(q.dtype);
        const int64_t col = pad256(n * es), icol = pad256(n * 4);
        const int64_t bin_bytes = 256 + 2 * icol + 5 * col + icol;   // run_mixed's scratch
        const int64_t host_bytes = mem == TPC_MPC_HOST ? 5 * col + 2 * icol : 0;   // v dy dphi front rear | hz iters
        StreamOrderScope order(h, s);
        rc = order.begin();
        if (rc) return rc;
        rc = ensure(h, &h->mix, &h->mix_bytes, bin_bytes + host_bytes);
        if (rc) return rc;
        char* b = (char*)h->mix;
        const void *d_v = v, *d_dy = delta_y, *d_dphi = delta_phi;
        void *d_front = steering_front, *d_rear = steering_rear;
        const int32_t* d_hz = horizons;
        int32_t* d_iters = iters;
        if (mem == TPC_MPC_HOST) {
            char* hb = b + bin_bytes;
            HIP_TRY(h, hipMemcpyAsync(hb, v, n * es, hipMemcpyHostToDevice, s));
            HIP_TRY(h, hipMemcpyAsync(hb + col, delta_y, n * es, hipMemcpyHostToDevice, s));
            HIP_TRY(h, hipMemcpyAsync(hb + 2 * col, delta_phi, n * es, hipMemcpyHostToDevice, s));
            HIP_TRY(h, hipMemcpyAsync(hb + 5 * col, horizons, n * 4, hipMemcpyHostToDevice, s));
            d_v = hb; d_dy = hb + col; d_dphi = hb + 2 * col; d_front = hb + 3 * col; d_rear = hb + 4 * col;
            d_hz = (const int32_t*)(hb + 5 * col);
            d_iters = iters ? (int32_t*)(hb + 5 * col + icol) : nullptr;
        }
        HIP_TRY(h, hipMemsetAsync(h->ws_words + 1, 0, sizeof(uint32_t), s));
        if (q.dtype == TPC_MPC_F64)
            rc = run_mixed<double>(h, &q, n, d_hz, (const double*)d_v, (const double*)d_dy, (const double*)d_dphi,
                                   (double*)d_front, (double*)d_rear, d_iters, b, s);
        else
            rc = run_mixed<float>(h, &q, n, d_hz, (const float*)d_v, (const float*)d_dy, (const float*)d_dphi,
                                  (float*)d_front, (float*)d_rear, d_iters, b, s);
        if (rc) return rc;
        if (mem == TPC_MPC_HOST) {
            HIP_TRY(h, hipMemcpyAsync(steering_front, d_front, n * es, hipMemcpyDeviceToHost, s));
            HIP_TRY(h, hipMemcpyAsync(steering_rear, d_rear, n * es, hipMemcpyDeviceToHost, s));
            if (iters) HIP_TRY(h, hipMemcpyAsync(iters, d_iters, n * 4, hipMemcpyDeviceToHost, s));
            HIP_TRY(h, hipStreamSynchronize(s));
        }
        rc = order.end();
        if (rc) return rc;
        return finish_flags(h, flags_out, s);
    });
}
