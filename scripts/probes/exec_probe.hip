// Microbenchmark: does a lone wavefront issue VALU instructions faster when only part of EXEC is set?
//   hipcc --offload-arch=gfx950 -O3 scripts/probes/exec_probe.hip -o gpurun_out/exec_probe && gpurun_out/exec_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int CHAINS>
__global__ void probe(double* out, uint64_t* clocks, int active_lanes, int iters) {
    const int lane = threadIdx.x;
    double a[CHAINS];
    for (int c = 0; c < CHAINS; ++c) a[c] = 1.0 + lane * 1e-3 + c;
    const double m = 0.999999, b = 1e-9;
    uint64_t t0 = 0, t1 = 0;
    if (lane < active_lanes) {
        t0 = clock64();
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
#pragma unroll
                for (int c = 0; c < CHAINS; ++c) a[c] = __builtin_fma(a[c], m, b);
        }
        t1 = clock64();
    }
    double s = 0;
    for (int c = 0; c < CHAINS; ++c) s += a[c];
    out[lane] = s;
    if (lane == 0) clocks[0] = t1 - t0;
}

int main() {
    double* out; uint64_t* clk;
    hipMalloc(&out, 64 * 8); hipMalloc(&clk, 8);
    const int iters = 4096;
    for (int lanes : {64, 32, 16, 8, 1}) {
        uint64_t c1 = 0, c4 = 0;
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL(probe<1>, dim3(1), dim3(64), 0, 0, out, clk, lanes, iters);
            hipMemcpy(&c1, clk, 8, hipMemcpyDeviceToHost);
            hipLaunchKernelGGL(probe<4>, dim3(1), dim3(64), 0, 0, out, clk, lanes, iters);
            hipMemcpy(&c4, clk, 8, hipMemcpyDeviceToHost);
        }
        printf("active lanes %2d: dependent chain %.2f clocks per v_fma_f64, 4 independent chains %.2f clocks per v_fma_f64\n", lanes,
               (double)c1 / (iters * 16.0), (double)c4 / (iters * 16.0 * 4));
    }
    return 0;
}
