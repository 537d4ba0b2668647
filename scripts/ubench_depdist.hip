// Microbenchmark: fp64 VALU issue cost versus dependency distance, 1 wave per SIMD on gfx950.
// CH independent chains are interleaved, so every instruction depends on the one CH slots back.
// build+run on the GPU box: hipcc --offload-arch=gfx950 -O3 scripts/ubench_depdist.hip -o /tmp/ub && /tmp/ub
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CH, int MODE>
__global__ __launch_bounds__(64) void k(double* out, int iters, double a, double b) {
    double x[8];
    for (int i = 0; i < 8; ++i) x[i] = a + threadIdx.x * 1e-9 + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 48; ++r) {
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                if (MODE == 0) asm volatile("v_add_f64 %0, %0, %1" : "+v"(x[c]) : "v"(b));
                if (MODE == 1) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x[c]) : "v"(b));
                if (MODE == 2) asm volatile("v_max_f64 %0, %0, %1" : "+v"(x[c]) : "v"(b));
                if (MODE == 3) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(x[c]) : "v"(b));
                if (MODE == 4) {   // alternate add / mul along the chain
                    if (r & 1) asm volatile("v_add_f64 %0, %0, %1" : "+v"(x[c]) : "v"(b));
                    else asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x[c]) : "v"(b));
                }
                if (MODE == 5) asm volatile("v_add_f64 %0, %0, %1" : "+v"(x[c]) : "s"(b));   // SGPR operand
            }
        }
    }
    double s = 0;
    for (int i = 0; i < 8; ++i) s += x[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}
template <int CH, int MODE> void run(const char* name, double* d, int blocks = 1024) {
    const int iters = 2000, per = 48 * CH;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<CH, MODE><<<blocks, 64>>>(d, 10, 1.0, 1.0000001);
    hipEventRecord(e0);
    k<CH, MODE><<<blocks, 64>>>(d, iters, 1.0, 1.0000001);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-12s distance=%d  ns/instr=%.3f\n", name, CH, ms * 1e6 / ((double)iters * per));
}
template <int MODE> void sweep(const char* name, double* d) {
    run<1, MODE>(name, d); run<2, MODE>(name, d); run<3, MODE>(name, d); run<4, MODE>(name, d); run<6, MODE>(name, d); run<8, MODE>(name, d);
}
int main() {
    double* d; hipMalloc(&d, 4096 * 64 * 8);
    sweep<0>("v_add_f64", d); sweep<1>("v_mul_f64", d); sweep<2>("v_max_f64", d); sweep<3>("v_fma_f64", d);
    sweep<4>("add/mul", d); sweep<5>("add sgpr", d);
    return 0;
}
