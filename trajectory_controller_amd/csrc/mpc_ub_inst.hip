// One translation unit per horizon (compile with -DTPC_UB_H=<H>): instantiates the LANE_FMA kernels
// (mpc_ub.h) for fp64 / fp32, compact model, and exports their launcher.
#include <cstdint>
#include <type_traits>

#include "mpc_ub.h"

#ifndef TPC_UB_H
#error "compile with -DTPC_UB_H=<horizon>"
#endif

namespace tpc {

hipError_t ub_pg_asm_launch(const CompactArgs& a, const Knobs& k, const Workspace& ws, int64_t need, hipStream_t s);
hipError_t ub_pg_asm_launch_h10(const CompactArgs& a, const Knobs& k, const Workspace& ws, int64_t need, hipStream_t s);

namespace {

constexpr int kH = TPC_UB_H;

// persistent-wave count of a PG kernel (see pg_grid in mpc_lane_inst.hip)
template <class Tag, class Kernel>
inline int ub_grid(Kernel kernel, int block) {
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
template <typename T, bool EQB, int MODE> struct TagPg {};

// coordinate descent + queue order: everything up to the projected-gradient launches (also the front half of the
// GROUP family, mpc_group_inst.hip, which consumes the same records)
template <typename T, bool EQB>
hipError_t phase1(const CompactArgs& a, const Knobs& k, const Workspace& ws, hipStream_t s) {
    T* recs = (T*)ws.state;
    hipError_t e = hipMemsetAsync(ws.ticket, 0, sizeof(uint32_t), s);
    if (e == hipSuccess) e = hipMemsetAsync(ws.stats, 0, 3 * sizeof(unsigned long long), s);
    if (e == hipSuccess) e = order_begin(ws.sort_temp, s);
    if (e != hipSuccess) return e;
    const int cd_grid = (int)((a.n + kWave - 1) / kWave);
    if (ws.ev) (void)hipEventRecord(ws.ev[0], s);
    hipLaunchKernelGGL((ub_cd_kernel<T, kH, EQB>), dim3(cd_grid), dim3(kWave), 0, s, a, k, recs, ws.keys, ws.rank,
                       (uint32_t*)ws.sort_temp, ws.stats);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    e = order_finish(ws.keys, ws.rank, ws.order, a.n, ws.sort_temp, s);
    if (e != hipSuccess) return e;
    if (ws.ev) (void)hipEventRecord(ws.ev[1], s);
    return hipSuccess;
}
// one build of the projected-gradient kernel (it returns at once unless the coordinate-descent kernel's screen picked it)
template <typename T, bool EQB, int MODE>
hipError_t pg_launch(const CompactArgs& a, const Knobs& k, const Workspace& ws, hipStream_t s) {
    constexpr int bt = kWave * UbPlan<T, kH>::occ;
    const int64_t need = (a.n + bt - 1) / bt;
#if (TPC_UB_H == 20 || TPC_UB_H == 10) && !defined(TPC_UB_NO_ASM)   // (TPC_UB_NO_ASM: the compiler's loop, for A/B runs -- scripts/build_ub_variant.sh)
    if constexpr (sizeof(T) == 8 && EQB && MODE == 2) {   // mpc_ub_asm.h: the same kernel written by hand (N = 20: 536 instructions per iteration instead of 708)
        const int64_t waves = (a.n + kWave - 1) / kWave;  // one wavefront per workgroup (its own translation unit: mpc_ub_asm_inst.hip)
        return kH == 20 ? ub_pg_asm_launch(a, k, ws, waves, s) : ub_pg_asm_launch_h10(a, k, ws, waves, s);
    }
#endif
    int cap = ub_grid<TagPg<T, EQB, MODE>>(ub_pg_kernel<T, kH, EQB, MODE>, bt);
    if (ws.max_waves > 0 && cap > ws.max_waves / UbPlan<T, kH>::occ) cap = ws.max_waves / UbPlan<T, kH>::occ;   // (a presolve holds the other SIMDs: tpc_mpc_api.cpp)
    hipLaunchKernelGGL((ub_pg_kernel<T, kH, EQB, MODE>), dim3((unsigned)(need < cap ? need : cap)), dim3(bt), 0, s, a, k,
                       (const T*)ws.state, (const uint32_t*)ws.order, ws.ticket, ws.stats, order_queue_len(ws.sort_temp));
    return hipGetLastError();
}

template <typename T, bool EQB>
hipError_t run(const CompactArgs& a, const Knobs& k, const Workspace& ws, hipStream_t s) {
    if (a.n <= 0) return hipSuccess;
    hipError_t e = phase1<T, EQB>(a, k, ws, s);
    if (e != hipSuccess) return e;
    // every build goes out; those the coordinate-descent kernel's screens did not pick return at once
    e = pg_launch<T, EQB, 2>(a, k, ws, s);
    if constexpr (sizeof(T) == 4) {
        if (e == hipSuccess) e = pg_launch<T, EQB, 1>(a, k, ws, s);
    }
    if (e == hipSuccess) e = pg_launch<T, EQB, 0>(a, k, ws, s);
    if (ws.ev) (void)hipEventRecord(ws.ev[2], s);
    return e;
}

}  // namespace

#define TPC_CAT2(a, b) a##b
#define TPC_CAT(a, b) TPC_CAT2(a, b)

// `equal_bounds`: both inputs share lower and upper (the host checked); other bounds take the build
// with the extra per-step addition.
hipError_t TPC_CAT(ub_compact_h, TPC_UB_H)(int dtype, int equal_bounds, const CompactArgs& a, const Knobs& k,
                                            const Workspace& ws, hipStream_t s) {
    if (dtype == 0) {
        // fp64 at N = 30 / 40: GROUP takes LANE_FMA's requests (tpc_mpc_api.cpp, pick_algo); the screened one-lane kernels
        // of those horizons -- the ones that parked their registers in scratch around every refill -- are not built
        if constexpr (kH >= 30) return hipErrorInvalidValue;
        else return equal_bounds ? run<double, true>(a, k, ws, s) : run<double, false>(a, k, ws, s);
    }
    return run<float, true>(a, k, ws, s);   // fp32 keeps dlib's coordinates: no bound-dependent build (mpc_ub_model.h)
}

// The two halves the GROUP family borrows (mpc_group_inst.hip): the coordinate-descent kernel with the queue order,
// and the exact-stop-test build of the projected-gradient kernel for a batch the screen refused.
hipError_t TPC_CAT(ub_phase1_h, TPC_UB_H)(int dtype, int equal_bounds, const CompactArgs& a, const Knobs& k,
                                           const Workspace& ws, hipStream_t s) {
    if (dtype == 0) return equal_bounds ? phase1<double, true>(a, k, ws, s) : phase1<double, false>(a, k, ws, s);
    return phase1<float, true>(a, k, ws, s);
}
hipError_t TPC_CAT(ub_exact_h, TPC_UB_H)(int dtype, int equal_bounds, const CompactArgs& a, const Knobs& k,
                                          const Workspace& ws, hipStream_t s) {
    if (dtype == 0) return equal_bounds ? pg_launch<double, true, 0>(a, k, ws, s) : pg_launch<double, false, 0>(a, k, ws, s);
    return pg_launch<float, true, 0>(a, k, ws, s);
}

}  // namespace tpc
