// Microbenchmark: what a dependency that crosses between the vector and the scalar unit costs one wave
// (gfx950).  Each round is a dependent chain x -> ... -> x; ns per round, one wave per SIMD.
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench_cross.hip -o /tmp/ubx && /tmp/ubx
#include <hip/hip_runtime.h>
#include <cstdio>
template <int P>
__global__ __launch_bounds__(64) void k(double* out, int iters, double a) {
    double x = a + threadIdx.x * 1e-9, y = 1.0 + 1e-12, e = 0.5;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            int lo, hi;
            if (P == 0)   // reference: two dependent DP ops
                asm volatile("v_add_f64 %0, %0, %1\n\tv_max_f64 %0, %0, %2" : "+v"(x) : "v"(y), "v"(e));
            if (P == 1) { // v_cmp -> VCC -> v_cndmask (all VALU)
                asm volatile("v_add_f64 %0, %0, %1\n\tv_cmp_ge_f64 vcc, %0, %2" : "+v"(x) : "v"(y), "v"(e) : "vcc");
                lo = __double2loint(x); hi = __double2hiint(x);
                asm volatile("v_cndmask_b32 %0, %0, %0, vcc\n\tv_cndmask_b32 %1, %1, %1, vcc" : "+v"(lo), "+v"(hi) :: "vcc");
                x = __hiloint2double(hi, lo);
            }
            if (P == 2) { // v_cmp -> SGPR pair -> v_cndmask (VALU-written SGPR read by VALU)
                asm volatile("v_add_f64 %0, %0, %1\n\tv_cmp_ge_f64 s[20:21], %0, %2" : "+v"(x) : "v"(y), "v"(e) : "s20", "s21");
                lo = __double2loint(x); hi = __double2hiint(x);
                asm volatile("s_nop 1\n\tv_cndmask_b32 %0, %0, %0, s[20:21]\n\tv_cndmask_b32 %1, %1, %1, s[20:21]" : "+v"(lo), "+v"(hi) :: "s20", "s21");
                x = __hiloint2double(hi, lo);
            }
            if (P == 3) { // v_cmp -> SGPR -> s_and_b64 -> v_cndmask (through the scalar unit)
                asm volatile("v_add_f64 %0, %0, %1\n\tv_cmp_ge_f64 s[20:21], %0, %2\n\ts_and_b64 s[20:21], s[20:21], exec" : "+v"(x) : "v"(y), "v"(e) : "s20", "s21", "scc");
                lo = __double2loint(x); hi = __double2hiint(x);
                asm volatile("v_cndmask_b32 %0, %0, %0, s[20:21]\n\tv_cndmask_b32 %1, %1, %1, s[20:21]" : "+v"(lo), "+v"(hi) :: "s20", "s21");
                x = __hiloint2double(hi, lo);
            }
            if (P == 4) { // v_readlane -> SGPR -> VALU operand
                asm volatile("v_add_f64 %0, %0, %1" : "+v"(x) : "v"(y));
                lo = __double2loint(x); hi = __double2hiint(x);
                asm volatile("v_readlane_b32 s20, %1, 7\n\tv_readlane_b32 s21, %2, 7\n\ts_nop 3\n\tv_max_f64 %0, %0, s[20:21]" : "+v"(x) : "v"(lo), "v"(hi) : "s20", "s21");
            }
            if (P == 5) { // v_readlane -> s_and_b32 -> VALU operand
                asm volatile("v_add_f64 %0, %0, %1" : "+v"(x) : "v"(y));
                lo = __double2loint(x); hi = __double2hiint(x);
                asm volatile("v_readlane_b32 s20, %1, 7\n\tv_readlane_b32 s21, %2, 7\n\ts_and_b32 s20, s20, 0xffffffc0\n\tv_max_f64 %0, %0, s[20:21]" : "+v"(x) : "v"(lo), "v"(hi) : "s20", "s21", "scc");
            }
            if (P == 6)   // v_mov_b64_dpp row_newbcast (VALU only)
                asm volatile("v_add_f64 %0, %0, %1\n\ts_nop 1\n\tv_mov_b64_dpp %0, %0 row_newbcast:7 row_mask:0xf bank_mask:0xf" : "+v"(x) : "v"(y));
            if (P == 7) { // v_cmp -> SGPR -> s_cmp_lg_u64 -> s_cselect -> v_cndmask (the PG verdict select)
                asm volatile("v_add_f64 %0, %0, %1\n\tv_cmp_ge_f64 s[20:21], %0, %2\n\ts_cmp_lg_u64 s[20:21], 0\n\ts_cselect_b64 s[20:21], -1, 0" : "+v"(x) : "v"(y), "v"(e) : "s20", "s21", "scc");
                lo = __double2loint(x); hi = __double2hiint(x);
                asm volatile("v_cndmask_b32 %0, %0, %0, s[20:21]\n\tv_cndmask_b32 %1, %1, %1, s[20:21]" : "+v"(lo), "+v"(hi) :: "s20", "s21");
                x = __hiloint2double(hi, lo);
            }
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = x;
}
template <int P> void run(const char* name, double* d) {
    const int iters = 20000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<P><<<256, 64>>>(d, 10, 1.0);
    (void)hipEventRecord(e0);
    k<P><<<256, 64>>>(d, iters, 1.0);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-64s %.1f ns per round (%.0f cycles at 2.39 GHz)\n", name, ms * 1e6 / (iters * 16.0), ms * 1e6 / (iters * 16.0) * 2.39);
}
int main() {
    double* d; (void)hipMalloc(&d, 256 * 64 * 8);
    run<0>("0: v_add -> v_max (two dependent DP ops)", d);
    run<1>("1: v_add -> v_cmp vcc -> v_cndmask vcc", d);
    run<2>("2: v_add -> v_cmp sgpr -> v_cndmask sgpr", d);
    run<3>("3: v_add -> v_cmp sgpr -> s_and_b64 -> v_cndmask", d);
    run<4>("4: v_add -> 2 v_readlane -> v_max with sgpr operand", d);
    run<5>("5: v_add -> 2 v_readlane -> s_and_b32 -> v_max with sgpr operand", d);
    run<6>("6: v_add -> v_mov_b64_dpp row_newbcast", d);
    run<7>("7: v_add -> v_cmp sgpr -> s_cmp -> s_cselect -> v_cndmask", d);
    return 0;
}
