// Microbenchmark: issue cost per SIMD of the fp32 multiply-add forms by waves per SIMD (gfx950):
//   v_fmac_f32 (VOP2, accumulates in place), v_fma_f32 (VOP3, three sources), v_mul / v_add, and the same mix the
//   LANE_FMA fp32 kernel has (2 v_fma : 1 v_fmac : 0.7 min/max/med3).
//   hipcc --offload-arch=gfx950 -O3 scripts/probes/fp32_rates.hip -o /tmp/fp32r && /tmp/fp32r
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ __launch_bounds__(64) void k(float* out, int iters, float a, float b) {
    float x[8], y[8];
    for (int i = 0; i < 8; ++i) { x[i] = a + threadIdx.x * 1e-6f + i; y[i] = x[i] * 0.5f; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 32; ++r) {
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                if (MODE == 0) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x[c]) : "v"(b), "v"(y[c]));
                if (MODE == 1) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x[c]) : "v"(b), "v"(y[c]));
                if (MODE == 2) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(x[c]) : "v"(b), "v"(y[c]), "v"(y[(c + 1) & 7]));
                if (MODE == 3) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[c]) : "v"(b));
                if (MODE == 4) asm volatile("v_fma_f32 %0, -%1, %2, %0" : "+v"(x[c]) : "v"(b), "v"(y[c]));
                if (MODE == 5) asm volatile("v_fma_f32 %0, %1, %2, %0 clamp" : "+v"(x[c]) : "v"(b), "v"(y[c]));
                if (MODE == 6) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[c]) : "v"(b));
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += x[i] + y[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}
template <int MODE> void run(const char* name, float* d) {
    printf("%-34s", name);
    for (int blocks : {1024, 2048, 4096}) {
        const int iters = 4000, per = 32 * 8;
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        k<MODE><<<blocks, 64>>>(d, 10, 1.0f, 1.0000001f);
        (void)hipEventRecord(e0);
        k<MODE><<<blocks, 64>>>(d, iters, 1.0f, 1.0000001f);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("  %.3f", ms * 1e6 / ((double)iters * per) / (blocks / 1024.0));
    }
    printf("   ns per instruction per SIMD at 1 / 2 / 4 waves per SIMD\n");
}
int main() {
    float* d; (void)hipMalloc(&d, 4096 * 64 * 4);
    run<6>("v_mul_f32 (VOP2)", d); run<3>("v_add_f32 (VOP2)", d); run<0>("v_fmac_f32 (VOP2)", d);
    run<1>("v_fma_f32 d = a*b + d (VOP3)", d); run<2>("v_fma_f32 d = a*b + c (VOP3)", d);
    run<4>("v_fma_f32 with neg modifier", d); run<5>("v_fma_f32 with clamp", d);
    return 0;
}
