// The hand-scheduled projected-gradient kernel of the headline workload (mpc_ub_asm.h) in a translation unit of its
// own: its register allocation problem (162 pinned VGPRs around one 1 100-instruction asm statement) wants its own
// compiler flags (csrc/Makefile).
#include <cstdint>
#include <type_traits>

#include "mpc_ub_asm.h"

namespace tpc {

namespace {
template <class K>
hipError_t launch(K kernel, int& cap, int& simds, const CompactArgs& a, const Knobs& k, const Workspace& ws, int64_t need, hipStream_t s) {
    if (cap == 0) {   // persistent grid: as many wavefronts as fit (N = 20: one per SIMD)
        int cus = 256, per_cu = 4;
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, kWave, 0) != hipSuccess || per_cu < 1) per_cu = 1;
        if (per_cu > 12) per_cu = 12;
        cap = cus * per_cu;
        simds = cus * 4;
    }
    int grid = cap;
    if (need < grid) grid = (int)need;
    if (ws.max_waves > 0 && grid > ws.max_waves) grid = ws.max_waves;   // (a presolve holds the other SIMDs: tpc_mpc_api.cpp)
    if (grid <= 0) return hipSuccess;                                   // (set-up only)
    hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(kWave), 0, s, ub_asm_args(a, k, ws));
    return hipGetLastError();
}
}  // namespace

hipError_t ub_pg_asm_launch(const CompactArgs& a, const Knobs& k, const Workspace& ws, int64_t need, hipStream_t s) {
    static int cap = 0, simds = 0;
    return launch(ub_pg_asm_kernel, cap, simds, a, k, ws, need, s);
}
// Several wavefronts on a SIMD share its issue slots: each runs that much slower, and a batch is not done before its longest
// instance is -- so the build that lets three of them in (they hide each other's refill passes) is for batches in which every
// lane has several instances to go through; below six per lane, one wavefront per SIMD (N = 10: 1 048 576 instances 2.24 ms
// with three per SIMD against 3.9 with one; 262 144: 0.92 with three).
hipError_t ub_pg_asm_launch_h10(const CompactArgs& a, const Knobs& k, const Workspace& ws, int64_t need, hipStream_t s) {
    static int cap_solo = 0, cap_multi = 0, simds_solo = 0, simds = 0;
    if (need / 6 > 1024 && cap_multi == 0) {   // (the first large batch: learn the device's SIMD count from the multi build's set-up)
        hipError_t e = launch(ub_pg_asm_kernel_h10<false>, cap_multi, simds, a, k, ws, 0, s);
        if (e != hipSuccess) return e;
    }
    if (simds == 0 || need / 6 <= simds) return launch(ub_pg_asm_kernel_h10<true>, cap_solo, simds_solo, a, k, ws, need, s);
    return launch(ub_pg_asm_kernel_h10<false>, cap_multi, simds, a, k, ws, need, s);
}

}  // namespace tpc
