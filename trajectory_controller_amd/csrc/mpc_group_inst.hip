// One translation unit per horizon (compile with -DTPC_GROUP_H=<H>): instantiates the GROUP kernels (mpc_group.h:
// G lanes per instance) for fp64 / fp32, compact model, at the group sizes that divide the horizon, and exports
// their launcher.  The coordinate-descent phase, the queue order and the exact-stop-test fallback are the LANE_FMA
// unit's of the same horizon (mpc_ub_inst.hip).
#include <cstdint>

#include "mpc_group.h"

#ifndef TPC_GROUP_H
#error "compile with -DTPC_GROUP_H=<horizon>"
#endif

namespace tpc {

#define TPC_CAT2(a, b) a##b
#define TPC_CAT(a, b) TPC_CAT2(a, b)
hipError_t TPC_CAT(ub_phase1_h, TPC_GROUP_H)(int, int, const CompactArgs&, const Knobs&, const Workspace&, hipStream_t);
hipError_t TPC_CAT(ub_exact_h, TPC_GROUP_H)(int, int, const CompactArgs&, const Knobs&, const Workspace&, hipStream_t);

namespace {

constexpr int kH = TPC_GROUP_H;

// compute units of the current device (cached: the query costs tens of microseconds)
inline int device_cus() {
    constexpr int kMaxDev = 64;
    static int cache[kMaxDev] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    if (dev >= 0 && dev < kMaxDev && cache[dev] > 0) return cache[dev];
    int cus = 256;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    if (dev >= 0 && dev < kMaxDev) cache[dev] = cus;
    return cus;
}
template <typename T, int G, bool EQB> struct Tag {};

// group sizes built per horizon (chunks of 3 .. 20 steps)
constexpr bool group_built(int H, int G) {
    return H == 10 ? (G == 2 || G == 4) : (H == 20 || H == 30 || H == 40) && (G == 2 || G == 4 || G == 8);
}

template <typename T, int G, bool EQB, bool MOVED = true>
hipError_t pg(const CompactArgs& a, const Knobs& k, const Workspace& ws, hipStream_t s) {
    if constexpr (!group_built(kH, G)) {
        return hipErrorInvalidValue;
    } else {
        constexpr int NG = GroupPlan<T, kH, G>::NG, OCC = GroupPlan<T, kH, G>::occ;
        const int64_t need = (a.n + NG - 1) / NG;
        // persistent grid: one wavefront per SIMD; as many as the kernel is built for (fp32: two) from the batch size at
        // which AUTO's table says the pair is ahead (Workspace::group_pair).  An explicit share of the chip
        // (tpc_mpc_x_set_group_share) is taken as given, up to what fits.
        int cap = device_cus() * 4 * (ws.group_pair ? OCC : 1);
        if (ws.max_waves > 0) cap = ws.max_waves < device_cus() * 4 * OCC ? ws.max_waves : device_cus() * 4 * OCC;
        hipLaunchKernelGGL((group_pg_kernel<T, kH, G, EQB, MOVED>), dim3((unsigned)(need < cap ? need : cap)), dim3(kWave), 0, s, a, k,
                           (const T*)ws.state, (const uint32_t*)ws.order, ws.ticket, ws.stats, order_queue_len(ws.sort_temp));
        return hipGetLastError();
    }
}
// fp32: both stop-test builds, back to back -- the coordinate-descent kernel's screens picked one (the other returns at its first load)
template <int G>
hipError_t pg_f32(const CompactArgs& a, const Knobs& k, const Workspace& ws, hipStream_t s) {
    const hipError_t e = pg<float, G, true, true>(a, k, ws, s);
    return e != hipSuccess ? e : pg<float, G, true, false>(a, k, ws, s);
}

template <typename T, bool EQB>
hipError_t pg_any(int G, const CompactArgs& a, const Knobs& k, const Workspace& ws, hipStream_t s) {
    switch (G) {
        case 2: return pg<T, 2, EQB>(a, k, ws, s);
        case 4: return pg<T, 4, EQB>(a, k, ws, s);
        case 8: return pg<T, 8, EQB>(a, k, ws, s);
    }
    return hipErrorInvalidValue;
}

}  // namespace

// G lanes per instance (2, 4 or 8: group_built).
hipError_t TPC_CAT(group_compact_h, TPC_GROUP_H)(int dtype, int equal_bounds, int G, const CompactArgs& a,
                                                  const Knobs& k, const Workspace& ws, hipStream_t s) {
    if (a.n <= 0) return hipSuccess;
    if (!group_built(kH, G)) return hipErrorInvalidValue;
    hipError_t e = TPC_CAT(ub_phase1_h, TPC_GROUP_H)(dtype, equal_bounds, a, k, ws, s);
    if (e != hipSuccess) return e;
    if (dtype == 0) e = equal_bounds ? pg_any<double, true>(G, a, k, ws, s) : pg_any<double, false>(G, a, k, ws, s);
    else e = G == 2 ? pg_f32<2>(a, k, ws, s) : (G == 4 ? pg_f32<4>(a, k, ws, s) : (G == 8 ? pg_f32<8>(a, k, ws, s) : hipErrorInvalidValue));
    if (e != hipSuccess) return e;
    // a batch the screen of the select-free stop test refused: LANE_FMA's exact build, on the same records
    e = TPC_CAT(ub_exact_h, TPC_GROUP_H)(dtype, equal_bounds, a, k, ws, s);
    if (ws.ev) (void)hipEventRecord(ws.ev[2], s);
    return e;
}

}  // namespace tpc
