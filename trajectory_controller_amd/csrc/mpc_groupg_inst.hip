// One translation unit per horizon (compile with -DTPC_GROUPG_H=<H>, H in 10, 20, 30, 40): instantiates the
// general-form GROUP kernels (mpc_groupg.h: G lanes per instance, per-instance A, B, C, Q, R, bounds, per-step
// targets) and exports their launcher.  The phase in front -- coordinate descent and the queue order -- and the fallback
// for a batch the stop-test screen refuses are borrowed:
//   cold starts at N = 10, 20 (fp64 and fp32)      the general-form LANE_FMA unit of the horizon (mpc_ubg_inst.hip);
//   the controller state in or out, N = 30, 40      the bit-exact LANE unit (mpc_lane_inst.hip; fp64 only): its
//                                                   coordinate-descent kernel applies the warm-start shift and leaves
//                                                   the records this kernel continues from, its unfused kernel keeps
//                                                   the state for a refused batch.
#include <cstdint>

#include "mpc_groupg.h"

#ifndef TPC_GROUPG_H
#error "compile with -DTPC_GROUPG_H=<horizon>"
#endif

namespace tpc {

#define TPC_CAT2(a, b) a##b
#define TPC_CAT(a, b) TPC_CAT2(a, b)
#if TPC_GROUPG_H <= 20
hipError_t TPC_CAT(ubg_phase1_h, TPC_GROUPG_H)(int, int, const GeneralArgs&, const Knobs&, const Workspace&, hipStream_t);
hipError_t TPC_CAT(ubg_exact_h, TPC_GROUPG_H)(int, int, const GeneralArgs&, const Knobs&, const Workspace&, hipStream_t);
#endif
hipError_t TPC_CAT(lane_general_phase1_h, TPC_GROUPG_H)(int, const GeneralArgs&, const Knobs&, const Workspace&, hipStream_t);
hipError_t TPC_CAT(lane_general_refused_h, TPC_GROUPG_H)(int, const GeneralArgs&, const Knobs&, const Workspace&, hipStream_t);

namespace {

constexpr int kH = TPC_GROUPG_H;

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

// group sizes built per horizon (chunks of 3 .. 10 steps: the general model keeps ~22 values per step in registers)
constexpr bool groupg_built(int H, int G) {
    return H == 10 ? (G == 2 || G == 4) : H == 20 ? (G == 2 || G == 4 || G == 8) : (H == 30 || H == 40) && (G == 4 || G == 8);
}

template <typename T, int I, int G, bool STATE>
hipError_t pg(const GeneralArgs& a, const Knobs& k, const Workspace& ws, hipStream_t s) {
    if constexpr (!groupg_built(kH, G)) {
        return hipErrorInvalidValue;
    } else {
        constexpr int NG = GroupPlan<T, kH, G>::NG;
        const int64_t need = (a.n + NG - 1) / NG;
        int cap = device_cus() * 4;   // one wavefront per SIMD
        if (ws.max_waves > 0 && ws.max_waves < cap) cap = ws.max_waves;
        hipLaunchKernelGGL((groupg_pg_kernel<T, I, kH, G, STATE>), dim3((unsigned)(need < cap ? need : cap)), dim3(kWave), 0, s, a, k,
                           (const T*)ws.state, (const uint32_t*)ws.order, ws.ticket, ws.stats, order_queue_len(ws.sort_temp));
        return hipGetLastError();
    }
}
template <typename T, int I, bool STATE>
hipError_t pg_any(int G, const GeneralArgs& a, const Knobs& k, const Workspace& ws, hipStream_t s) {
    switch (G) {
        case 2: return pg<T, I, 2, STATE>(a, k, ws, s);
        case 4: return pg<T, I, 4, STATE>(a, k, ws, s);
        case 8: return pg<T, I, 8, STATE>(a, k, ws, s);
    }
    return hipErrorInvalidValue;
}

}  // namespace

// G lanes per instance (groupg_built).
hipError_t TPC_CAT(groupg_general_h, TPC_GROUPG_H)(int dtype, int inputs, int G, const GeneralArgs& a, const Knobs& k,
                                                    const Workspace& ws, hipStream_t s) {
    if (a.n <= 0) return hipSuccess;
    if (!groupg_built(kH, G) || (inputs != 1 && inputs != 2)) return hipErrorInvalidValue;
    const bool state = a.controls != nullptr || a.v != nullptr;
    const bool via_lane = state || kH > 20;
    if (via_lane && dtype != 0) return hipErrorInvalidValue;   // (the host routes those to LANE)
    hipError_t e;
    if (via_lane) {
        e = TPC_CAT(lane_general_phase1_h, TPC_GROUPG_H)(inputs, a, k, ws, s);
        if (e != hipSuccess) return e;
        if (state) e = inputs == 1 ? pg_any<double, 1, true>(G, a, k, ws, s) : pg_any<double, 2, true>(G, a, k, ws, s);
        else e = inputs == 1 ? pg_any<double, 1, false>(G, a, k, ws, s) : pg_any<double, 2, false>(G, a, k, ws, s);
        if (e != hipSuccess) return e;
        e = TPC_CAT(lane_general_refused_h, TPC_GROUPG_H)(inputs, a, k, ws, s);   // a batch the screen refused
    } else {
#if TPC_GROUPG_H <= 20
        e = TPC_CAT(ubg_phase1_h, TPC_GROUPG_H)(dtype, inputs, a, k, ws, s);
        if (e != hipSuccess) return e;
        if (dtype == 0) e = inputs == 1 ? pg_any<double, 1, false>(G, a, k, ws, s) : pg_any<double, 2, false>(G, a, k, ws, s);
        else e = inputs == 1 ? pg_any<float, 1, false>(G, a, k, ws, s) : pg_any<float, 2, false>(G, a, k, ws, s);
        if (e != hipSuccess) return e;
        e = TPC_CAT(ubg_exact_h, TPC_GROUPG_H)(dtype, inputs, a, k, ws, s);      // a batch the screen refused
#else
        e = hipErrorInvalidValue;
#endif
    }
    if (ws.ev) (void)hipEventRecord(ws.ev[2], s);
    return e;
}

}  // namespace tpc
