// Microbenchmark: one wave's projected-gradient step, tail variants (what do off-chain instructions cost?)
#include <hip/hip_runtime.h>
#include <cstdio>
#define FM(acc, k, q) "v_fmac_f64_dpp %" #acc ", %2, %" #k " row_newbcast:" #q " row_mask:0xf bank_mask:0xf\n\t"
template <int V>
__global__ __launch_bounds__(64) void k(double* out, const double* in, int iters) {
    double kq[8];
    for (int i = 0; i < 8; ++i) kq[i] = in[i * 64 + threadIdx.x] * 1e-3;
    double u = in[600 + threadIdx.x] * 0.1, v = u, g = in[700 + threadIdx.x];
    const double lo = -0.38, hi = 0.38, il = 1e-3, beta = 0.9, glo = 1e300, ghi = 1e300;
    double csum = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            double a0 = g, a1 = 0;
            asm volatile("s_nop 1\n\t" FM(0,3,0) FM(1,4,1) FM(0,5,2) FM(1,6,3) FM(0,7,4) FM(1,8,5) FM(0,9,6) FM(1,10,7)
                         : "+v"(a0), "+v"(a1) : "v"(u), "v"(kq[0]), "v"(kq[1]), "v"(kq[2]), "v"(kq[3]), "v"(kq[4]), "v"(kq[5]), "v"(kq[6]), "v"(kq[7]));
            double df, t, vn, un, c = 0, d;
            if (V == 0) {
                asm volatile("v_add_f64 %0, %4, %5\n\tv_fma_f64 %1, -%6, %0, %7\n\tv_min_f64 %1, %1, %8\n\tv_max_f64 %2, %1, %9\n\t"
                             "v_add_f64 %1, %2, -%10\n\tv_fma_f64 %1, %11, %1, %2\n\tv_min_f64 %1, %1, %8\n\tv_max_f64 %3, %1, %9"
                             : "=&v"(df), "=&v"(t), "=&v"(vn), "=&v"(un) : "v"(a0), "v"(a1), "v"(il), "v"(u), "v"(hi), "v"(lo), "v"(v), "v"(beta));
            } else if (V == 1) {   // c chain right after df (what a compiler tends to do)
                asm volatile("v_add_f64 %0, %4, %5\n\tv_min_f64 %12, %0, %13\n\tv_max_f64 %12, %12, -%14\n\tv_fma_f64 %1, -%6, %0, %7\n\tv_min_f64 %1, %1, %8\n\tv_max_f64 %2, %1, %9\n\t"
                             "v_add_f64 %1, %2, -%10\n\tv_fma_f64 %1, %11, %1, %2\n\tv_min_f64 %1, %1, %8\n\tv_max_f64 %3, %1, %9"
                             : "=&v"(df), "=&v"(t), "=&v"(vn), "=&v"(un) : "v"(a0), "v"(a1), "v"(il), "v"(u), "v"(hi), "v"(lo), "v"(v), "v"(beta), "v"(c), "v"(glo), "v"(ghi));
            } else if (V == 2) {   // c chain at the very end
                asm volatile("v_add_f64 %0, %4, %5\n\tv_fma_f64 %1, -%6, %0, %7\n\tv_min_f64 %1, %1, %8\n\tv_max_f64 %2, %1, %9\n\t"
                             "v_add_f64 %1, %2, -%10\n\tv_fma_f64 %1, %11, %1, %2\n\tv_min_f64 %1, %1, %8\n\tv_max_f64 %3, %1, %9\n\t"
                             "v_min_f64 %12, %0, %13\n\tv_max_f64 %12, %12, -%14"
                             : "=&v"(df), "=&v"(t), "=&v"(vn), "=&v"(un), "=&v"(d) : "v"(a0), "v"(a1), "v"(il), "v"(u), "v"(hi), "v"(lo), "v"(v), "v"(beta), "v"(c), "v"(glo), "v"(ghi));
            } else if (V == 3) {   // interleaved
                asm volatile("v_add_f64 %0, %4, %5\n\tv_fma_f64 %1, -%6, %0, %7\n\tv_min_f64 %12, %0, %13\n\tv_min_f64 %1, %1, %8\n\tv_max_f64 %12, %12, -%14\n\tv_max_f64 %2, %1, %9\n\t"
                             "v_add_f64 %1, %2, -%10\n\tv_fma_f64 %1, %11, %1, %2\n\tv_min_f64 %1, %1, %8\n\tv_max_f64 %3, %1, %9"
                             : "=&v"(df), "=&v"(t), "=&v"(vn), "=&v"(un) : "v"(a0), "v"(a1), "v"(il), "v"(u), "v"(hi), "v"(lo), "v"(v), "v"(beta), "v"(c), "v"(glo), "v"(ghi));
            } else if (V == 4) {   // V0 + 2 independent adds
                asm volatile("v_add_f64 %0, %4, %5\n\tv_fma_f64 %1, -%6, %0, %7\n\tv_add_f64 %12, %13, %14\n\tv_min_f64 %1, %1, %8\n\tv_add_f64 %12, %13, %14\n\tv_max_f64 %2, %1, %9\n\t"
                             "v_add_f64 %1, %2, -%10\n\tv_fma_f64 %1, %11, %1, %2\n\tv_min_f64 %1, %1, %8\n\tv_max_f64 %3, %1, %9"
                             : "=&v"(df), "=&v"(t), "=&v"(vn), "=&v"(un) : "v"(a0), "v"(a1), "v"(il), "v"(u), "v"(hi), "v"(lo), "v"(v), "v"(beta), "v"(c), "v"(glo), "v"(ghi));
            } else if (V == 5) {   // V3 + stop verdict as select (cmp -> s_cmp -> s_cselect -> cndmask x4)
                unsigned long long m, ga;
                asm volatile("v_add_f64 %0, %4, %5\n\tv_fma_f64 %1, -%6, %0, %7\n\tv_min_f64 %12, %0, %13\n\tv_min_f64 %1, %1, %8\n\tv_max_f64 %12, %12, -%14\n\tv_max_f64 %2, %1, %9\n\t"
                             "v_add_f64 %1, %2, -%10\n\tv_fma_f64 %1, %11, %1, %2\n\tv_min_f64 %1, %1, %8\n\tv_max_f64 %3, %1, %9"
                             : "=&v"(df), "=&v"(t), "=&v"(vn), "=&v"(un) : "v"(a0), "v"(a1), "v"(il), "v"(u), "v"(hi), "v"(lo), "v"(v), "v"(beta), "v"(c), "v"(glo), "v"(ghi));
                asm volatile("v_cmp_ge_f64 %0, |%2|, %3\n\ts_cmp_lg_u64 %0, 0\n\ts_cselect_b64 %1, -1, 0" : "=&s"(m), "=&s"(ga) : "v"(c), "v"(1e-30) : "scc");
                int l0, h0, l1, h1;
                asm volatile("v_cndmask_b32 %0, %4, %5, %8\n\tv_cndmask_b32 %1, %6, %7, %8" : "=&v"(l0), "=&v"(h0), "=&v"(l1), "=&v"(h1)
                             : "v"(__double2loint(u)), "v"(__double2loint(un)), "v"(__double2hiint(u)), "v"(__double2hiint(un)), "s"(ga));
                un = __hiloint2double(h0, l0);
            }
            if (V >= 6) {   // V3 + N extra 32-bit VALU instructions on a side value
                asm volatile("v_add_f64 %0, %4, %5\n\tv_fma_f64 %1, -%6, %0, %7\n\tv_min_f64 %12, %0, %13\n\tv_min_f64 %1, %1, %8\n\tv_max_f64 %12, %12, -%14\n\tv_max_f64 %2, %1, %9\n\t"
                             "v_add_f64 %1, %2, -%10\n\tv_fma_f64 %1, %11, %1, %2\n\tv_min_f64 %1, %1, %8\n\tv_max_f64 %3, %1, %9"
                             : "=&v"(df), "=&v"(t), "=&v"(vn), "=&v"(un) : "v"(a0), "v"(a1), "v"(il), "v"(u), "v"(hi), "v"(lo), "v"(v), "v"(beta), "v"(c), "v"(glo), "v"(ghi));
                int w = it;
                if (V == 6) asm volatile("v_and_b32 %0, 0x7fffffff, %0" : "+v"(w));
                if (V == 7) asm volatile("v_and_b32 %0, 0x7fffffff, %0\n\tv_and_b32 %0, 0x7fffffff, %0\n\tv_and_b32 %0, 0x7fffffff, %0\n\tv_and_b32 %0, 0x7fffffff, %0" : "+v"(w));
                if (V == 8) asm volatile("v_mov_b32 %0, %0\n\tv_mov_b32 %0, %0\n\tv_mov_b32 %0, %0\n\tv_mov_b32 %0, %0" : "+v"(w));
                if (V == 9) asm volatile("v_cndmask_b32 %0, %0, %0, vcc\n\tv_cndmask_b32 %0, %0, %0, vcc\n\tv_cndmask_b32 %0, %0, %0, vcc\n\tv_cndmask_b32 %0, %0, %0, vcc" : "+v"(w));
                if (V == 10) { double z = u; asm volatile("v_mov_b64 %0, %0\n\tv_mov_b64 %0, %0\n\tv_mov_b64 %0, %0\n\tv_mov_b64 %0, %0" : "+v"(z)); csum += z; }
                if (V == 11) { double z = u; asm volatile("v_add_f64 %0, %0, %0\n\tv_add_f64 %0, %0, %0\n\tv_add_f64 %0, %0, %0\n\tv_add_f64 %0, %0, %0" : "+v"(z)); csum += z; }
                csum += w;
            }
            csum += c;
            v = vn; u = un;
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = u + v + csum;
}
template <int V> void run(const char* name, double* d, const double* in) {
    for (int blocks : {256, 4096}) {
        const int iters = 5000;
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        k<V><<<blocks, 64>>>(d, in, 10);
        (void)hipEventRecord(e0);
        k<V><<<blocks, 64>>>(d, in, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        const double per = ms * 1e6 / (iters * 4.0);
        printf("%-44s blocks=%4d  %.1f ns/iter per wave, %.1f ns per SIMD\n", name, blocks, per, per / (blocks < 1024 ? 1 : blocks / 1024.0));
    }
}
int main() {
    double *d, *in; (void)hipMalloc(&d, 4096 * 64 * 8); (void)hipMalloc(&in, 1024 * 8);
    double h[1024]; for (int i = 0; i < 1024; ++i) h[i] = 0.001 * (i % 97) + 0.01;
    (void)hipMemcpy(in, h, sizeof h, hipMemcpyHostToDevice);
    run<0>("0: fmacs + u chain", d, in);
    run<1>("1: + c chain right after df", d, in);
    run<2>("2: + c chain at the end", d, in);
    run<3>("3: + c chain interleaved", d, in);
    run<4>("4: + 2 independent adds", d, in);
    run<5>("5: interleaved + verdict select", d, in);
    run<6>("6: V3 + 1 v_and_b32", d, in);
    run<7>("7: V3 + 4 v_and_b32 (dependent)", d, in);
    run<8>("8: V3 + 4 v_mov_b32", d, in);
    run<9>("9: V3 + 4 v_cndmask_b32", d, in);
    run<10>("10: V3 + 4 v_mov_b64", d, in);
    run<11>("11: V3 + 4 v_add_f64 (dependent)", d, in);
    return 0;
}
