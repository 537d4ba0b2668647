// The hand-scheduled projected-gradient kernel of the headline workload (mpc_ub_asm.h) in a translation unit of its
// own: its register allocation problem (162 pinned VGPRs around one 1 100-instruction asm statement) wants its own
// compiler flags (csrc/Makefile).
#include <cstdint>
#include <type_traits>

#include "mpc_ub_asm.h"

namespace tpc {

hipError_t ub_pg_asm_launch(const CompactArgs& a, const Knobs& k, const Workspace& ws, int64_t need, hipStream_t s) {
    static int cap = 0;   // persistent grid: one wavefront per SIMD
    if (cap == 0) {
        int cus = 256, per_cu = 4;
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, ub_pg_asm_kernel, kWave, 0) != hipSuccess || per_cu < 1) per_cu = 1;
        if (per_cu > 4) per_cu = 4;
        cap = cus * per_cu;
    }
    hipLaunchKernelGGL(ub_pg_asm_kernel, dim3((unsigned)(need < cap ? need : cap)), dim3(kWave), 0, s, ub_asm_args(a, k, ws));
    return hipGetLastError();
}

}  // namespace tpc
