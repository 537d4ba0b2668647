// Microbenchmark: issue cost per SIMD of fp64 VALU forms by waves per SIMD (gfx950): does the number of distinct VGPR
// sources matter (register-file ports), do SGPR operands or the accumulate form help?
//   hipcc --offload-arch=gfx950 -O3 scripts/probes/fp64_rates.hip -o /tmp/fp64r && /tmp/fp64r
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ __launch_bounds__(64) void k(double* out, int iters, double a, double b) {
    double x[8], y[8], z[8];
    for (int i = 0; i < 8; ++i) { x[i] = a + threadIdx.x * 1e-9 + i; y[i] = x[i] * 0.5; z[i] = x[i] * 0.25; }
    double vb = b + threadIdx.x * 1e-12;   // a per-lane (VGPR) copy of b
    int w[8];
    for (int i = 0; i < 8; ++i) w[i] = threadIdx.x + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                if (MODE == 0) asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(x[c]) : "v"(vb), "v"(y[c]));                 // d += v*v
                if (MODE == 1) asm volatile("v_fma_f64 %0, %1, %2, %3" : "=v"(x[c]) : "v"(vb), "v"(y[c]), "v"(z[c]));   // d = v*v + v (3 distinct)
                if (MODE == 2) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(x[c]) : "s"(b), "v"(y[c]));               // d = s*v + d
                if (MODE == 3) asm volatile("v_fma_f64 %0, %1, %2, %3" : "=v"(x[c]) : "s"(b), "v"(y[c]), "v"(z[c]));    // d = s*v + v
                if (MODE == 4) asm volatile("v_add_f64 %0, %1, %2" : "=v"(x[c]) : "v"(y[c]), "v"(z[c]));                // d = v + v
                if (MODE == 5) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x[c]) : "v"(vb));                             // d = d*v
                if (MODE == 6) asm volatile("v_fma_f64 %0, %1, %2, %3 clamp" : "=v"(x[c]) : "v"(vb), "v"(y[c]), "v"(z[c]));
                if (MODE == 7) asm volatile("v_max_f64 %0, %1, %2" : "=v"(x[c]) : "v"(y[c]), "v"(z[c]));
                if (MODE == 8) asm volatile("v_fma_f64 %0, %1, %1, %2" : "=v"(x[c]) : "v"(y[c]), "v"(z[c]));            // d = v*v(same) + v
                if (MODE == 9) asm volatile("v_mov_b64 %0, %1" : "=v"(x[c]) : "v"(y[c]));
                if (MODE == 10) asm volatile("v_accvgpr_write_b32 a0, %0\n v_accvgpr_read_b32 %0, a0" : "+v"(w[c]) : : "a0");
            }
        }
    }
    double s = 0;
    for (int i = 0; i < 8; ++i) s += x[i] + y[i] + z[i] + w[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}
template <int MODE> void run(const char* name, double* d, int per_mult = 1) {
    printf("%-40s", name);
    for (int blocks : {1024, 2048}) {
        const int iters = 3000, per = 16 * 8 * per_mult;
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        k<MODE><<<blocks, 64>>>(d, 10, 1.0, 1.0000001);
        (void)hipEventRecord(e0);
        k<MODE><<<blocks, 64>>>(d, iters, 1.0, 1.0000001);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("  %.3f", ms * 1e6 / ((double)iters * per) / (blocks / 1024.0));
    }
    printf("   ns per instruction per SIMD at 1 / 2 waves per SIMD\n");
}
int main() {
    double* d; (void)hipMalloc(&d, 4096 * 64 * 8);
    run<0>("v_fmac_f64  d += v*v", d); run<1>("v_fma_f64   d = v*v + v (3 distinct)", d);
    run<2>("v_fma_f64   d = s*v + d", d); run<3>("v_fma_f64   d = s*v + v", d); run<8>("v_fma_f64   d = v*v(same) + v", d);
    run<6>("v_fma_f64   3 distinct, clamp", d); run<4>("v_add_f64   d = v + v", d); run<5>("v_mul_f64   d = d*v", d);
    run<7>("v_max_f64   d = max(v, v)", d); run<9>("v_mov_b64", d); run<10>("accvgpr write + read (2 instr)", d, 2);
    return 0;
}
