// One translation unit per horizon (compile with -DTPC_UBG_H=<H>, H in 4, 5, 10, 20): instantiates the LANE_FMA
// kernels of the general model (mpc_ubg.h) for fp64 / fp32 and one or two inputs, and exports their launcher.
#include <cstdint>
#include <type_traits>

#include "mpc_ubg.h"

#ifndef TPC_UBG_H
#error "compile with -DTPC_UBG_H=<horizon>"
#endif

namespace tpc {

namespace {

constexpr int kH = TPC_UBG_H;

// persistent-wave count of a PG kernel (see pg_grid in mpc_lane_inst.hip)
template <class Tag, class Kernel>
inline int ubg_grid(Kernel kernel, int block) {
    constexpr int kMaxDev = 64;
    static int cache[kMaxDev] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    if (dev >= 0 && dev < kMaxDev && cache[dev] > 0) return cache[dev];
    int cus = 256, per_cu = 4;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, block, 0) != hipSuccess || per_cu < 1)
        per_cu = 1;
    if (per_cu > 4) per_cu = 4;
    if (dev >= 0 && dev < kMaxDev) cache[dev] = cus * per_cu;
    return cus * per_cu;
}
template <typename T, int I> struct TagFast {};
template <typename T, int I> struct TagExact {};

// coordinate descent + queue order (also the front half of the general-form GROUP kernels, mpc_groupg_inst.hip)
template <typename T, int I>
hipError_t phase1(const GeneralArgs& a, const Knobs& k, const Workspace& ws, hipStream_t s) {
    T* recs = (T*)ws.state;
    hipError_t e = hipMemsetAsync(ws.ticket, 0, sizeof(uint32_t), s);
    if (e == hipSuccess) e = hipMemsetAsync(ws.stats, 0, 3 * sizeof(unsigned long long), s);
    if (e == hipSuccess) e = order_begin(ws.sort_temp, s);
    if (e != hipSuccess) return e;
    const int cd_grid = (int)((a.n + kWave - 1) / kWave);
    if (ws.ev) (void)hipEventRecord(ws.ev[0], s);
    hipLaunchKernelGGL((ubg_cd_kernel<T, I, kH>), dim3(cd_grid), dim3(kWave), 0, s, a, k, recs, ws.keys, ws.rank,
                       (uint32_t*)ws.sort_temp, ws.stats);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    e = order_finish(ws.keys, ws.rank, ws.order, a.n, ws.sort_temp, s);
    if (e != hipSuccess) return e;
    if (ws.ev) (void)hipEventRecord(ws.ev[1], s);
    return hipSuccess;
}
template <typename T, int I, bool FAST>
hipError_t pg_launch(const GeneralArgs& a, const Knobs& k, const Workspace& ws, hipStream_t s) {
    constexpr int bt = kWave * UbgPlan<T, kH>::occ;
    const int64_t need = (a.n + bt - 1) / bt;
    const int cap = ubg_grid<std::conditional_t<FAST, TagFast<T, I>, TagExact<T, I>>>(ubg_pg_kernel<T, I, kH, FAST>, bt);
    hipLaunchKernelGGL((ubg_pg_kernel<T, I, kH, FAST>), dim3((unsigned)(need < cap ? need : cap)), dim3(bt), 0, s, a, k,
                       (const T*)ws.state, (const uint32_t*)ws.order, ws.ticket, ws.stats, order_queue_len(ws.sort_temp));
    return hipGetLastError();
}

template <typename T, int I>
hipError_t run(const GeneralArgs& a, const Knobs& k, const Workspace& ws, hipStream_t s) {
    if (a.n <= 0) return hipSuccess;
    if (a.controls || a.v) return hipErrorInvalidValue;   // cold starts only (the host routes the rest to LANE)
    hipError_t e = phase1<T, I>(a, k, ws, s);
    if (e != hipSuccess) return e;
    // both builds go out; the one the coordinate-descent kernel's screen did not pick returns at once
    e = pg_launch<T, I, true>(a, k, ws, s);
    if (e == hipSuccess) e = pg_launch<T, I, false>(a, k, ws, s);
    if (ws.ev) (void)hipEventRecord(ws.ev[2], s);
    return e;
}

}  // namespace

#define TPC_CAT2(a, b) a##b
#define TPC_CAT(a, b) TPC_CAT2(a, b)

hipError_t TPC_CAT(ub_general_h, TPC_UBG_H)(int dtype, int inputs, const GeneralArgs& a, const Knobs& k,
                                             const Workspace& ws, hipStream_t s) {
    if (inputs != 1 && inputs != 2) return hipErrorInvalidValue;
    if (dtype == 0) return inputs == 1 ? run<double, 1>(a, k, ws, s) : run<double, 2>(a, k, ws, s);
    return inputs == 1 ? run<float, 1>(a, k, ws, s) : run<float, 2>(a, k, ws, s);
}

// the two halves the general-form GROUP kernels borrow (mpc_groupg_inst.hip)
hipError_t TPC_CAT(ubg_phase1_h, TPC_UBG_H)(int dtype, int inputs, const GeneralArgs& a, const Knobs& k, const Workspace& ws, hipStream_t s) {
    if (dtype == 0) return inputs == 1 ? phase1<double, 1>(a, k, ws, s) : phase1<double, 2>(a, k, ws, s);
    return inputs == 1 ? phase1<float, 1>(a, k, ws, s) : phase1<float, 2>(a, k, ws, s);
}
hipError_t TPC_CAT(ubg_exact_h, TPC_UBG_H)(int dtype, int inputs, const GeneralArgs& a, const Knobs& k, const Workspace& ws, hipStream_t s) {
    if (dtype == 0) return inputs == 1 ? pg_launch<double, 1, false>(a, k, ws, s) : pg_launch<double, 2, false>(a, k, ws, s);
    return inputs == 1 ? pg_launch<float, 1, false>(a, k, ws, s) : pg_launch<float, 2, false>(a, k, ws, s);
}

}  // namespace tpc
