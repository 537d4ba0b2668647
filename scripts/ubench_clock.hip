// Microbenchmark: what the fp64 vector pipe delivers and at which clock (gfx950).
// Per configuration: ns per v_add_f64 wave-instruction per SIMD (aggregate over the waves of a SIMD),
// the in-kernel shader clock = delta s_memtime / delta s_memrealtime x 100 MHz (MI355X_MICROARCH.md,
// DVFS give-back item 6; median over workgroups), hence cycles per instruction.
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench_clock.hip -o /tmp/ubc && /tmp/ubc
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
template <int CH>
__global__ __launch_bounds__(64) void k(double* out, unsigned long long* stamps, int iters, double a, double b) {
    double x[8];
    for (int i = 0; i < 8; ++i) x[i] = a + threadIdx.x * 1e-9 + i;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 32; ++r)
#pragma unroll
            for (int c = 0; c < CH; ++c) asm volatile("v_add_f64 %0, %0, %1" : "+v"(x[c]) : "v"(b));
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    double s = 0;
    for (int i = 0; i < 8; ++i) s += x[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = c1 - c0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}
template <int CH> void run(double* d, unsigned long long* st, int blocks) {
    const int iters = 20000, per = 32 * CH;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) k<CH><<<blocks, 64>>>(d, st, iters, 1.0, 1.0000001);   // warm: let the clock settle
    hipEventRecord(e0);
    k<CH><<<blocks, 64>>>(d, st, iters, 1.0, 1.0000001);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(2 * blocks);
    hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> ghz(blocks);
    for (int i = 0; i < blocks; ++i) ghz[i] = (double)h[2 * i] / (double)h[2 * i + 1] * 0.1;
    std::sort(ghz.begin(), ghz.end());
    const double clk = ghz[blocks / 2];
    const double waves_per_simd = blocks / 1024.0;
    const double ns = ms * 1e6 / ((double)iters * per) / (waves_per_simd < 1 ? 1 : waves_per_simd);
    printf("waves/SIMD %.0f, %d independent chains: %.3f ns per wave-instruction per SIMD, clock %.3f GHz -> %.2f cycles/instr; "
           "chip fp64 add rate %.1f Tinstr-lanes/s\n", waves_per_simd, CH, ns, clk, ns * clk, 1024 * 64 / ns / 1e3);
}
int main() {
    double* d; hipMalloc(&d, 8192 * 64 * 8);
    unsigned long long* st; hipMalloc(&st, 8192 * 2 * 8);
    for (int b : {1024, 2048, 4096}) { run<1>(d, st, b); run<4>(d, st, b); run<8>(d, st, b); }
    return 0;
}
