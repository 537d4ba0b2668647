// Microbenchmark: fp32 VALU issue cost vs waves per SIMD on gfx950 (companion of ubench_dp.hip).
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CH, int MODE>
__global__ __launch_bounds__(64) void k(float* out, int iters, float a, float b) {
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = a + threadIdx.x * 1e-3f + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 32; ++r) {
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                if (MODE == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[c]) : "v"(b));
                if (MODE == 1) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[c]) : "v"(b));
                if (MODE == 2) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(*(double*)&x[2 * (c & 3)]) : "v"((double)b));
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += x[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}
template <int CH, int MODE> void run(const char* name, float* d, int blocks) {
    const int iters = 2000, per = 32 * CH;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<CH, MODE><<<blocks, 64>>>(d, 10, 1.0f, 1.0000001f);
    hipEventRecord(e0);
    k<CH, MODE><<<blocks, 64>>>(d, iters, 1.0f, 1.0000001f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-24s blocks=%d chains=%d  ns/instr/wave=%.3f  aggregate ns/instr/SIMD=%.3f\n", name, blocks, CH,
           ms * 1e6 / ((double)iters * per), ms * 1e6 / ((double)iters * per) / (blocks > 1024 ? blocks / 1024.0 : 1.0));
}
int main() {
    float* d; hipMalloc(&d, 8192 * 64 * 4);
    for (int b : {1024, 2048, 4096, 8192}) run<4, 0>("v_add_f32", d, b);
    for (int b : {1024, 2048, 4096}) run<1, 0>("v_add_f32 dependent", d, b);
    for (int b : {1024, 2048, 4096}) run<4, 1>("v_fma_f32", d, b);
    for (int b : {1024, 2048, 4096}) run<4, 2>("v_pk_add_f32", d, b);
    return 0;
}
