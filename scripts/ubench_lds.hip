// Microbenchmark: exposed LDS latency behind fp64 VALU work, 1 wave per SIMD (4 per CU) on gfx950.
// Each iteration: R x ds_read2st64_b64 (+ optional ds_write2st64_b64), N independent v_add_f64,
// s_waitcnt lgkmcnt(0), one use of the data.  time(N) - time without LDS = what the LDS ops cost.
// build+run on the GPU box: hipcc --offload-arch=gfx950 -O3 scripts/ubench_lds.hip -o /tmp/ub && /tmp/ub
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double2_t __attribute__((ext_vector_type(2)));
template <int N, int R, int W>
__global__ __launch_bounds__(64) void k(double* out, int iters, double b) {
    __shared__ double s[80][64];
    for (int i = 0; i < 80; ++i) s[i][threadIdx.x] = i + threadIdx.x;
    double x[8];
    for (int i = 0; i < 8; ++i) x[i] = 1.0 + threadIdx.x * 1e-9 + i;
    const unsigned addr = threadIdx.x * 8;
    double acc = 0;
    for (int it = 0; it < iters; ++it) {
        double2_t d[4];
        if (R >= 1) asm volatile("ds_read2st64_b64 %0, %1 offset0:0 offset1:1" : "=v"(d[0]) : "v"(addr));
        if (R >= 2) asm volatile("ds_read2st64_b64 %0, %1 offset0:2 offset1:3" : "=v"(d[1]) : "v"(addr));
        if (R >= 3) asm volatile("ds_read2st64_b64 %0, %1 offset0:4 offset1:5" : "=v"(d[2]) : "v"(addr));
        if (R >= 4) asm volatile("ds_read2st64_b64 %0, %1 offset0:6 offset1:7" : "=v"(d[3]) : "v"(addr));
        if (W >= 1) asm volatile("ds_write2st64_b64 %0, %1, %2 offset0:10 offset1:11" : : "v"(addr), "v"(x[0]), "v"(x[1]));
#pragma unroll
        for (int r = 0; r < N; ++r) asm volatile("v_add_f64 %0, %0, %1" : "+v"(x[r & 7]) : "v"(b));
        if (R + W > 0) asm volatile("s_waitcnt lgkmcnt(0)");
#pragma unroll
        for (int r = 0; r < R; ++r) asm volatile("v_add_f64 %0, %0, %1" : "+v"(acc) : "v"(d[r].x));
    }
    double t = acc;
    for (int i = 0; i < 8; ++i) t += x[i];
    out[blockIdx.x * 64 + threadIdx.x] = t;
}
template <int N, int R, int W> double run(double* d, int blocks = 1024) {
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<N, R, W><<<blocks, 64>>>(d, 100, 1.0000001);
    hipEventRecord(e0);
    k<N, R, W><<<blocks, 64>>>(d, iters, 1.0000001);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e6 / iters;   // ns per iteration
}
template <int N> void row(double* d) {
    const double base = run<N, 0, 0>(d), r1 = run<N, 1, 0>(d), r2 = run<N, 2, 0>(d), r4 = run<N, 4, 0>(d), r2w = run<N, 2, 1>(d);
    printf("N=%3d valu: base %7.1f ns | +1 read %6.1f | +2 reads %6.1f | +4 reads %6.1f | +2 reads +1 write %6.1f   (extra ns per iteration)\n",
           N, base, r1 - base, r2 - base, r4 - base, r2w - base);
}
int main() {
    double* d; hipMalloc(&d, 4096 * 64 * 8);
    row<8>(d); row<16>(d); row<24>(d); row<32>(d); row<48>(d); row<64>(d); row<96>(d); row<128>(d);
    printf("one wave per CU (256 blocks):\n");
    {
        const double base = run<48, 0, 0>(d, 256), r2w = run<48, 2, 1>(d, 256);
        printf("N= 48 base %7.1f  +2 reads +1 write %6.1f\n", base, r2w - base);
    }
    return 0;
}
