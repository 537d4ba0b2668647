// Microbenchmark: issue cost of fp32 VALU instructions, plain and packed, by waves per SIMD (gfx950).
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench_pk.hip -o /tmp/ubpk && /tmp/ubpk
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float pkf __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(64) void k(float* out, int iters, float a, float b) {
    float x[8]; pkf p[8];
    for (int i = 0; i < 8; ++i) { x[i] = a + threadIdx.x * 1e-6f + i; p[i] = pkf{x[i], x[i] + 1.0f}; }
    pkf pb = {b, b};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 32; ++r) {
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                if (MODE == 0) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[c]) : "v"(b));
                if (MODE == 1) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[c]) : "v"(pb));
                if (MODE == 2) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(x[c]) : "v"(b), "v"(a));
                if (MODE == 3) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x[c]) : "v"(b), "v"(a));
                if (MODE == 4) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[c]) : "v"(pb));
                if (MODE == 5) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[c]) : "v"(pb));
                if (MODE == 6) asm volatile("v_min_f32 %0, %0, %1" : "+v"(x[c]) : "v"(b));
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += x[i] + p[i].x + p[i].y;
    out[blockIdx.x * 64 + threadIdx.x] = s;
}
template <int MODE> void run(const char* name, float* d) {
    for (int blocks : {1024, 2048, 4096}) {
        const int iters = 4000, per = 32 * 8;
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        k<MODE><<<blocks, 64>>>(d, 10, 1.0f, 1.0000001f);
        (void)hipEventRecord(e0);
        k<MODE><<<blocks, 64>>>(d, iters, 1.0f, 1.0000001f);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        const double ns = ms * 1e6 / ((double)iters * per) / (blocks / 1024.0);
        printf("%-16s %d waves/SIMD: %.3f ns per instruction per SIMD\n", name, blocks / 1024, ns);
    }
}
int main() {
    float* d; (void)hipMalloc(&d, 4096 * 64 * 4);
    run<0>("v_mul_f32", d); run<6>("v_min_f32", d); run<2>("v_med3_f32", d); run<3>("v_max3_f32", d);
    run<1>("v_pk_mul_f32", d); run<5>("v_pk_add_f32", d); run<4>("v_pk_fma_f32", d);
    return 0;
}
