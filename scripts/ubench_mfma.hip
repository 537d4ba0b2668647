// Can the matrix pipe do a plain IEEE fp64 add with one operand taken from an AGPR?
//   v_mfma_f64_4x4x4_4b_f64 D, A, B, C   (4 blocks of 4x4x4, one double per lane for A, B, C, D)
// With A a 0/1 pattern (identity: lane = i + 4*block + 16*k hot iff i == k) and the data in B,
// D = 1*B + 0*(other lanes' B) + C lands in the lane the data came from (D lane = j + 4*block + 16*i,
// B lane = j + 4*block + 16*k); with the roles swapped it does not.  This program (1) finds the lane layout /
// the B pattern for which D[lane] = A[lane] + C[lane], (2) checks bit-exactness against v_add_f64
// on random operands, (3) measures what the MFMA costs in issue slots next to fp64 VALU work.
// build+run on the GPU box: hipcc --offload-arch=gfx950 -O3 scripts/ubench_mfma.hip -o /tmp/ub && /tmp/ub
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
#include <random>

__global__ void probe(const double* a, const double* b, const double* c, double* d) {
    const int l = threadIdx.x;
    d[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(b[l], a[l], c[l], 0, 0, 0);   // pattern as A, data as B
}

__global__ void exact(const double* a, const double* c, const double* bpat, double* d_mfma, double* d_add, int n) {
    const int l = threadIdx.x;
    const double b = bpat[l];
    for (int i = blockIdx.x; i < n; i += gridDim.x) {
        const double x = a[(size_t)i * 64 + l], y = c[(size_t)i * 64 + l];
        d_mfma[(size_t)i * 64 + l] = __builtin_amdgcn_mfma_f64_4x4x4f64(b, x, y, 0, 0, 0);
        double s;
        asm volatile("v_add_f64 %0, %1, %2" : "=v"(s) : "v"(x), "v"(y));
        d_add[(size_t)i * 64 + l] = s;
    }
}

// N v_add_f64 on 8 independent chains + M MFMAs whose A operand sits in an AGPR, per iteration
template <int N, int M, bool DEP>
__global__ __launch_bounds__(64) void mix(double* out, int iters, double bb, const double* bpat) {
    double x[8];
    for (int i = 0; i < 8; ++i) x[i] = 1.0 + threadIdx.x * 1e-9 + i;
    const double b = bpat[threadIdx.x];
    double pm;
    asm volatile("v_accvgpr_write_b32 a0, %0\n v_accvgpr_write_b32 a1, %1" : : "v"(__double2loint(bb)), "v"(__double2hiint(bb)) : "a0", "a1");
    double acc[4] = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < N; ++r) {
            asm volatile("v_add_f64 %0, %0, %1" : "+v"(x[r & 7]) : "v"(bb));
            if (M > 0 && (r % (N / (M > 0 ? M : 1))) == 0 && r / (N / (M > 0 ? M : 1)) < M) {
                const int q = (r / (N / M)) & 3;
                if (DEP) asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, a[0:1], %0" : "+v"(acc[q]) : "v"(b));
                else asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, a[0:1], %2" : "=v"(acc[q]) : "v"(b), "v"(x[7]));
            }
        }
    }
    double t = acc[0] + acc[1] + acc[2] + acc[3];
    for (int i = 0; i < 8; ++i) t += x[i];
    out[blockIdx.x * 64 + threadIdx.x] = t;
}
template <int N, int M, bool DEP> double run(double* d, const double* bpat) {
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    mix<N, M, DEP><<<1024, 64>>>(d, 100, 1.0000001, bpat);
    hipEventRecord(e0);
    mix<N, M, DEP><<<1024, 64>>>(d, iters, 1.0000001, bpat);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e6 / iters;
}

int main() {
    double *a, *b, *c, *d;
    hipMallocManaged(&a, 64 * 8); hipMallocManaged(&b, 64 * 8); hipMallocManaged(&c, 64 * 8); hipMallocManaged(&d, 64 * 8);
    // (1) layout: one-hot B at lane p, A = 100 + lane: which D lanes light up and with whose A?
    int a_of[64][64];   // a_of[p][L] = lane whose A appears in D[L] when B is hot at p (or -1)
    for (int p = 0; p < 64; ++p) {
        for (int l = 0; l < 64; ++l) { a[l] = 100 + l; b[l] = (l == p); c[l] = 0; }
        probe<<<1, 64>>>(a, b, c, d); hipDeviceSynchronize();
        for (int l = 0; l < 64; ++l) a_of[p][l] = d[l] == 0 ? -1 : (int)(d[l] - 100);
    }
    // B pattern: lane p must be hot iff it contributes A[L] to D[L] for some L
    std::vector<double> pat(64, 0.0);
    int covered = 0, wrong = 0;
    for (int p = 0; p < 64; ++p) {
        bool self = false, other = false;
        for (int l = 0; l < 64; ++l) if (a_of[p][l] >= 0) { if (a_of[p][l] == l) self = true; else other = true; }
        if (self && !other) { pat[p] = 1.0; for (int l = 0; l < 64; ++l) covered += a_of[p][l] == l; }
        if (self && other) ++wrong;
    }
    printf("layout: hot lanes:"); for (int p = 0; p < 64; ++p) if (pat[p] != 0) printf(" %d", p);
    printf("\n  D lanes that receive their own A: %d of 64; ambiguous B lanes: %d\n", covered, wrong);
    printf("  B hot at lane 0 lights D lanes:"); for (int l = 0; l < 64; ++l) if (a_of[0][l] >= 0) printf(" %d<-A%d", l, a_of[0][l]); printf("\n");
    printf("  B hot at lane 5 lights D lanes:"); for (int l = 0; l < 64; ++l) if (a_of[5][l] >= 0) printf(" %d<-A%d", l, a_of[5][l]); printf("\n");
    if (covered != 64) { printf("no per-lane pattern\n"); return 0; }

    // (2) exactness
    const int n = 1 << 16;
    double *A, *C, *DM, *DA, *P;
    hipMallocManaged(&A, (size_t)n * 64 * 8); hipMallocManaged(&C, (size_t)n * 64 * 8);
    hipMallocManaged(&DM, (size_t)n * 64 * 8); hipMallocManaged(&DA, (size_t)n * 64 * 8); hipMallocManaged(&P, 64 * 8);
    for (int l = 0; l < 64; ++l) P[l] = pat[l];
    std::mt19937_64 rng(12345);
    for (size_t i = 0; i < (size_t)n * 64; ++i) {
        uint64_t ra = rng(), rc = rng();
        const int mode = (i / 64) % 8;
        double x, y;
        if (mode < 4) {   // random sign/exponent near each other so that rounding and cancellation happen
            const int e = 1023 + (int)(ra % 40) - 20;
            uint64_t xa = (ra & 0x800fffffffffffffull) | ((uint64_t)e << 52);
            uint64_t xc = (rc & 0x800fffffffffffffull) | ((uint64_t)(e + (int)(rc >> 60) - 8) << 52);
            memcpy(&x, &xa, 8); memcpy(&y, &xc, 8);
        } else if (mode < 6) {   // denormals and tiny values
            uint64_t xa = ra & 0x801fffffffffffffull, xc = rc & 0x803fffffffffffffull;
            memcpy(&x, &xa, 8); memcpy(&y, &xc, 8);
        } else if (mode == 6) {  // exact cancellation and zeros of both signs
            uint64_t xa = (ra & 0x800fffffffffffffull) | (1000ull << 52); memcpy(&x, &xa, 8);
            y = (i & 1) ? -x : x; if ((i & 6) == 2) x = (ra >> 63) ? -0.0 : 0.0;
        } else {                 // huge
            uint64_t xa = (ra & 0x800fffffffffffffull) | (2040ull << 52), xc = (rc & 0x800fffffffffffffull) | (2044ull << 52);
            memcpy(&x, &xa, 8); memcpy(&y, &xc, 8);
        }
        A[i] = x; C[i] = y;
    }
    exact<<<1024, 64>>>(A, C, P, DM, DA, n); hipDeviceSynchronize();
    size_t bad = 0, badzero = 0;
    for (size_t i = 0; i < (size_t)n * 64; ++i) {
        uint64_t u, v; memcpy(&u, &DM[i], 8); memcpy(&v, &DA[i], 8);
        if (u != v) { if (DM[i] == 0 && DA[i] == 0) ++badzero; else { if (bad < 5) printf("  mismatch: %a + %a = mfma %a, add %a\n", A[i], C[i], DM[i], DA[i]); ++bad; } }
    }
    printf("exactness: %zu operand pairs, %zu value mismatches, %zu sign-of-zero mismatches\n", (size_t)n * 64, bad, badzero);

    // (3) issue cost
    double* o; hipMalloc(&o, 1024 * 64 * 8);
    const double base = run<64, 0, false>(o, P);
    printf("64 v_add_f64 per iteration: %.1f ns\n", base);
    printf("  + 8  independent MFMA: %+.1f ns  (%.2f ns each)\n", run<64, 8, false>(o, P) - base, (run<64, 8, false>(o, P) - base) / 8);
    printf("  + 16 independent MFMA: %+.1f ns  (%.2f ns each)\n", run<64, 16, false>(o, P) - base, (run<64, 16, false>(o, P) - base) / 16);
    printf("  + 32 independent MFMA: %+.1f ns  (%.2f ns each)\n", run<64, 32, false>(o, P) - base, (run<64, 32, false>(o, P) - base) / 32);
    printf("  + 8  chained MFMA (4 chains): %+.1f ns\n", run<64, 8, true>(o, P) - base);
    printf("  + 16 chained MFMA (4 chains): %+.1f ns\n", run<64, 16, true>(o, P) - base);
    return 0;
}
