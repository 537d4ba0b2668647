// Microbenchmark: issue cost of fp64 VALU chains with 1 wave per SIMD on gfx950.
// build+run on the GPU box: hipcc --offload-arch=gfx950 -O3 scripts/ubench_dp.hip -o /tmp/ub && /tmp/ub
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CH, int MODE>
__global__ __launch_bounds__(64) void k(double* out, int iters, double a, double b) {
    double x[8];
    for (int i = 0; i < 8; ++i) x[i] = a + threadIdx.x * 1e-9 + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 32; ++r) {
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                if (MODE == 0) asm volatile("v_add_f64 %0, %0, %1" : "+v"(x[c]) : "v"(b));
                if (MODE == 1) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x[c]) : "v"(b));
                if (MODE == 2) asm volatile("v_max_f64 %0, %0, %1" : "+v"(x[c]) : "v"(b));
                if (MODE == 3) { asm volatile("v_add_f64 %0, %0, %1\n v_mov_b32 %2, %2" : "+v"(x[c]) : "v"(b), "v"(iters)); }
                if (MODE == 4) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(x[c]) : "v"(b));
            }
        }
    }
    double s = 0;
    for (int i = 0; i < 8; ++i) s += x[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}
template <int CH, int MODE> void run(const char* name, double* d, int blocks = 1024) {
    const int iters = 2000, per = 32 * CH;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<CH, MODE><<<blocks, 64>>>(d, 10, 1.0, 1.0000001);
    hipEventRecord(e0);
    k<CH, MODE><<<blocks, 64>>>(d, iters, 1.0, 1.0000001);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s blocks=%d chains=%d  ns/instr=%.3f  (cycles@2.4GHz %.2f)\n", name, blocks, CH, ms * 1e6 / ((double)iters * per), ms * 1e6 / ((double)iters * per) * 2.4);
}
int main() {
    double* d; hipMalloc(&d, 4096 * 64 * 8);
    run<1, 0>("v_add_f64 dependent", d); run<2, 0>("v_add_f64", d); run<4, 0>("v_add_f64", d); run<8, 0>("v_add_f64", d);
    run<1, 1>("v_mul_f64 dependent", d); run<4, 1>("v_mul_f64", d);
    run<1, 2>("v_max_f64 dependent", d); run<4, 2>("v_max_f64", d);
    run<1, 4>("v_fma_f64 dependent", d); run<4, 4>("v_fma_f64", d);
    run<1, 3>("add_f64 + mov_b32 (dep)", d); run<4, 3>("add_f64 + mov_b32", d);
    for (int b : {256, 512, 1024, 2048, 4096}) { run<4, 0>("v_add_f64 x waves", d, b); }
    for (int b : {1024, 2048, 4096}) { run<1, 0>("v_add_f64 dep x waves", d, b); }
    for (int b : {1024, 2048, 4096}) { run<4, 3>("add+mov x waves", d, b); }
    return 0;
}
