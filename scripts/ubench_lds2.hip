// Microbenchmark: what one LDS instruction costs a lone wave per SIMD when its latency is fully
// covered by independent fp64 VALU work (gfx950).  Variants by width / form.
// build+run on the GPU box: hipcc --offload-arch=gfx950 -O3 scripts/ubench_lds2.hip -o /tmp/ub && /tmp/ub
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double2_t __attribute__((ext_vector_type(2)));
typedef float float4_t __attribute__((ext_vector_type(4)));
template <int MODE, int K>
__global__ __launch_bounds__(64) void k(double* out, int iters, double b) {
    __shared__ double s[160][64];
    for (int i = 0; i < 160; ++i) s[i][threadIdx.x] = i + threadIdx.x;
    double x[8];
    for (int i = 0; i < 8; ++i) x[i] = 1.0 + threadIdx.x * 1e-9 + i;
    const unsigned a8 = threadIdx.x * 8, a16 = threadIdx.x * 16, a4 = threadIdx.x * 4;
    double acc = 0;
    for (int it = 0; it < iters; ++it) {
        double2_t d[K]; double e[K]; float f[K]; float4_t q[K];
#pragma unroll
        for (int r = 0; r < K; ++r) {
            if (MODE == 1) asm volatile("ds_read_b32 %0, %1" : "=v"(f[r]) : "v"(a4));
            if (MODE == 2) asm volatile("ds_read_b64 %0, %1" : "=v"(e[r]) : "v"(a8));
            if (MODE == 3) asm volatile("ds_read2st64_b64 %0, %1 offset0:0 offset1:1" : "=v"(d[r]) : "v"(a8));
            if (MODE == 4) asm volatile("ds_read_b128 %0, %1" : "=v"(q[r]) : "v"(a16));
            if (MODE == 5) asm volatile("ds_write_b64 %0, %1 offset:8192" : : "v"(a8), "v"(x[r & 7]));
            if (MODE == 6) asm volatile("ds_write2st64_b64 %0, %1, %2 offset0:20 offset1:21" : : "v"(a8), "v"(x[r & 7]), "v"(x[(r + 1) & 7]));
            if (MODE == 7) asm volatile("ds_write_b128 %0, %1 offset:16384" : : "v"(a16), "v"(q[0]));
            if (MODE == 8) asm volatile("ds_read2_b32 %0, %1 offset0:0 offset1:64" : "=v"(e[r]) : "v"(a4));
#pragma unroll
            for (int j = 0; j < 16; ++j) asm volatile("v_add_f64 %0, %0, %1" : "+v"(x[j & 7]) : "v"(b));
        }
#pragma unroll
        for (int j = 0; j < 64; ++j) asm volatile("v_add_f64 %0, %0, %1" : "+v"(x[j & 7]) : "v"(b));
        if (MODE > 0) asm volatile("s_waitcnt lgkmcnt(0)");
#pragma unroll
        for (int r = 0; r < K; ++r) {
            if (MODE == 1) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[0]) : "v"(f[r]));
            if (MODE == 2 || MODE == 8) asm volatile("v_add_f64 %0, %0, %1" : "+v"(acc) : "v"(e[r]));
            if (MODE == 3) asm volatile("v_add_f64 %0, %0, %1" : "+v"(acc) : "v"(d[r].x));
            if (MODE == 4) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[0]) : "v"(q[r].x));
        }
        if (MODE == 1 || MODE == 4) acc += f[0];
    }
    double t = acc;
    for (int i = 0; i < 8; ++i) t += x[i];
    out[blockIdx.x * 64 + threadIdx.x] = t;
}
template <int MODE, int K> double run(double* d) {
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE, K><<<1024, 64>>>(d, 100, 1.0000001);
    hipEventRecord(e0);
    k<MODE, K><<<1024, 64>>>(d, iters, 1.0000001);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e6 / iters;
}
template <int MODE> void row(const char* name, double* d) {
    const double b4 = run<0, 4>(d), b8 = run<0, 8>(d);
    const double t4 = run<MODE, 4>(d), t8 = run<MODE, 8>(d);
    printf("%-22s 4 per iteration: %+6.1f ns (%.2f each)   8 per iteration: %+6.1f ns (%.2f each)\n", name, t4 - b4, (t4 - b4) / 4, t8 - b8, (t8 - b8) / 8);
}
int main() {
    double* d; hipMalloc(&d, 4096 * 64 * 8);
    printf("base: 4 groups %.1f ns, 8 groups %.1f ns (16 v_add_f64 per group + 64)\n", run<0, 4>(d), run<0, 8>(d));
    row<1>("ds_read_b32", d); row<8>("ds_read2_b32", d); row<2>("ds_read_b64", d); row<3>("ds_read2st64_b64", d); row<4>("ds_read_b128", d);
    row<5>("ds_write_b64", d); row<6>("ds_write2st64_b64", d); row<7>("ds_write_b128", d);
    return 0;
}
