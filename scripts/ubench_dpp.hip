// Microbenchmark: v_fmac_f64_dpp (row_newbcast) vs plain fma and vs two v_mov_dpp + fma; permlane swaps.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CH, int MODE>
__global__ __launch_bounds__(64) void k(double* out, int iters, double a, double b) {
    double x[8];
    for (int i = 0; i < 8; ++i) x[i] = a + threadIdx.x * 1e-9 + i;
    double u = b + threadIdx.x * 1e-12, kk = 1e-9;
    unsigned p = threadIdx.x, q = threadIdx.x * 3;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 32; ++r) {
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                if (MODE == 0) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(x[c]) : "v"(u), "v"(kk));
                if (MODE == 1) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(x[c]) : "v"(u), "v"(kk));
                if (MODE == 2) { int tl, th; asm volatile("v_mov_b32_dpp %0, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %3 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "=&v"(tl), "=&v"(th) : "v"(__double2loint(u)), "v"(__double2hiint(u))); double t = __hiloint2double(th, tl); asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(x[c]) : "v"(t), "v"(kk)); }
                if (MODE == 3) asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "=v"(x[c]) : "v"(u));
                if (MODE == 4) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(p), "+v"(q));
                if (MODE == 5) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(p), "+v"(q));
                if (MODE == 6) asm volatile("v_mov_b32_dpp %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "=v"(p) : "v"(q));
                if (MODE == 7) asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(x[c]) : "v"(u), "v"(kk));
            }
        }
    }
    double s = p + q;
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
    double* d; (void)hipMalloc(&d, 4096 * 64 * 8);
    run<1, 0>("v_fma_f64 dependent", d); run<4, 0>("v_fma_f64", d);
    run<1, 7>("v_fmac_f64 dependent", d); run<4, 7>("v_fmac_f64", d);
    run<1, 1>("v_fmac_f64_dpp dependent", d); run<4, 1>("v_fmac_f64_dpp", d); run<8, 1>("v_fmac_f64_dpp", d);
    run<1, 2>("2 mov_dpp + fma dependent", d); run<4, 2>("2 mov_dpp + fma", d);
    run<4, 3>("v_mov_b64_dpp", d);
    run<1, 4>("permlane16_swap", d); run<1, 5>("permlane32_swap", d); run<1, 6>("v_mov_b32_dpp", d);
    run<4, 1>("v_fmac_f64_dpp x4096", d, 4096); run<4, 0>("v_fma_f64 x4096", d, 4096); run<4, 2>("2mov+fma x4096", d, 4096);
    return 0;
}
