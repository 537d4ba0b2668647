// Microbenchmark: does a long straight-line loop body (no reuse inside ~12 KB of code) limit issue
// when two waves share a SIMD?  Independent v_add_f32 / v_add_f64 streams, body of BODY instructions.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int BODY, int DP>
__global__ __launch_bounds__(64) void k(float* out, int iters, float a, float b) {
    float x[8]; double y[8];
    for (int i = 0; i < 8; ++i) { x[i] = a + threadIdx.x * 1e-3f + i; y[i] = x[i]; }
    double bd = b;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < BODY / 8; ++r) {
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                if (DP) asm volatile("v_add_f64 %0, %0, %1" : "+v"(y[c]) : "v"(bd));
                else asm volatile("v_add_f32_e64 %0, %0, %1" : "+v"(x[c]) : "v"(b));
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += x[i] + (float)y[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}
template <int BODY, int DP> void run(float* d, int blocks) {
    const int iters = 200000 / BODY * 16;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<BODY, DP><<<blocks, 64>>>(d, 2, 1.0f, 1.0000001f);
    hipEventRecord(e0);
    k<BODY, DP><<<blocks, 64>>>(d, iters, 1.0f, 1.0000001f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double per = ms * 1e6 / ((double)iters * BODY);
    printf("%s body=%5d instr (%5d B) blocks=%d  ns/instr/wave=%.3f  aggregate/SIMD=%.3f\n", DP ? "f64" : "f32", BODY,
           BODY * 8, blocks, per, per / (blocks / 1024.0));
}
int main() {
    float* d; hipMalloc(&d, 8192 * 64 * 4);
    for (int b : {1024, 2048}) { run<128, 0>(d, b); run<1536, 0>(d, b); run<4096, 0>(d, b); run<8192, 0>(d, b); }
    for (int b : {1024, 2048}) { run<128, 1>(d, b); run<1536, 1>(d, b); run<4096, 1>(d, b); }
    return 0;
}
