// One translation unit per horizon (compile with -DTPC_LANE_H=<H>): instantiates the LANE kernels
// for fp64/fp32, compact and general (I = 1, 2) models, and exports their launchers.
#include "mpc_lane.h"

#ifndef TPC_LANE_H
#error "compile with -DTPC_LANE_H=<horizon>"
#endif

namespace tpc {

namespace {

constexpr int kH = TPC_LANE_H;

// Persistent-wave count of the PG kernel: one wave per SIMD (the kernels need > 256 VGPRs).
inline int pg_grid(int64_t n) {
    int dev = 0;
    hipDeviceProp_t prop;
    int waves = 1024;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
        waves = prop.multiProcessorCount * 4;
    const int64_t need = (n + kWave - 1) / kWave;
    return (int)(need < waves ? need : waves);
}

// The fused PG kernel publishes controls[0] only; a caller that wants the controller state back
// (warm-start chains, tpc_mpc_rollout) gets the kernel that keeps it.
inline bool wants_state(const CompactArgs&) { return false; }
inline bool wants_state(const GeneralArgs& a) { return a.controls != nullptr || a.v != nullptr; }

template <typename T, int I, class Model, class Args>
hipError_t run(const Args& a, const Knobs& k, const Workspace& ws, hipStream_t s) {
    if (a.n <= 0) return hipSuccess;
    T* recs = (T*)ws.state;
    hipError_t e = hipMemsetAsync(ws.ticket, 0, sizeof(uint32_t), s);
    if (e == hipSuccess) e = hipMemsetAsync(ws.stats, 0, 2 * sizeof(unsigned long long), s);
    if (e != hipSuccess) return e;
    const int cd_grid = (int)((a.n + kWave - 1) / kWave);
    if (ws.ev) (void)hipEventRecord(ws.ev[0], s);
    hipLaunchKernelGGL((lane_cd_kernel<T, I, kH, Model, Args>), dim3(cd_grid), dim3(kWave), 0, s, a, k, recs,
                       ws.keys);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    e = order_desc(ws.keys, ws.order, a.n, ws.sort_temp, s);
    if (e != hipSuccess) return e;
    if (ws.ev) (void)hipEventRecord(ws.ev[1], s);
    if (wants_state(a))
        hipLaunchKernelGGL((lane_pg_kernel<T, I, kH, Model, Args>), dim3(pg_grid(a.n)), dim3(kWave), 0, s, a, k,
                           (const T*)recs, (const uint32_t*)ws.order, ws.ticket, ws.stats);
    else
        hipLaunchKernelGGL((lane_pg_fused_kernel<T, I, kH, Model, Args>), dim3(pg_grid(a.n)), dim3(kWave), 0, s,
                           a, k, (const T*)recs, (const uint32_t*)ws.order, ws.ticket, ws.stats);
    e = hipGetLastError();
    if (ws.ev) (void)hipEventRecord(ws.ev[2], s);
    return e;
}

}  // namespace

#define TPC_CAT2(a, b) a##b
#define TPC_CAT(a, b) TPC_CAT2(a, b)

int64_t TPC_CAT(lane_rec_len_h, TPC_LANE_H)(int dtype) {
    return dtype == 0 ? LaneRec<double, kH>::kLen : LaneRec<float, kH>::kLen;
}

hipError_t TPC_CAT(lane_compact_h, TPC_LANE_H)(int dtype, const CompactArgs& a, const Knobs& k,
                                                 const Workspace& ws, hipStream_t s) {
    if (dtype == 0) return run<double, 2, CompactModel<double>, CompactArgs>(a, k, ws, s);
    return run<float, 2, CompactModel<float>, CompactArgs>(a, k, ws, s);
}

hipError_t TPC_CAT(lane_general_h, TPC_LANE_H)(int dtype, int I, const GeneralArgs& a, const Knobs& k,
                                                 const Workspace& ws, hipStream_t s) {
    if (dtype == 0) {
        if (I == 2) return run<double, 2, GeneralModel<double, 2>, GeneralArgs>(a, k, ws, s);
        return run<double, 1, GeneralModel<double, 1>, GeneralArgs>(a, k, ws, s);
    }
    if (I == 2) return run<float, 2, GeneralModel<float, 2>, GeneralArgs>(a, k, ws, s);
    return run<float, 1, GeneralModel<float, 1>, GeneralArgs>(a, k, ws, s);
}

}  // namespace tpc
