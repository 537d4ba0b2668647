/* examples/sharded_inprocess.c -- INTEGRATION.md section 3, compiled: ONE host process (the reference host is
 * one C++ process, src/interface.cpp:3) drives G GPUs through the C ABI -- one handle and one stream per GPU,
 * one RCCL communicator, one tpc_mpc_solve_batch_compact_sharded call per GPU and batch between
 * tpc_mpc_group_begin / tpc_mpc_group_end -- and checks that every GPU ends up holding the control outputs of
 * ALL instances, bit-identical to one GPU solving the whole batch.
 *
 *   gcc -std=c99 -D__HIP_PLATFORM_AMD__ -Iinclude -I/opt/rocm/include examples/sharded_inprocess.c \
 *       -Ltrajectory_controller_amd/lib -ltpc_mpc -L/opt/rocm/lib -lamdhip64 \
 *       -Wl,-rpath,$PWD/trajectory_controller_amd/lib -o sharded_inprocess
 *   ./sharded_inprocess [G = all devices] [n_total = 100003] [horizon = 20]
 *
 * With G = 1 the library would skip the exchange (a world of one); the example then asks for a real one-rank
 * communicator (tpc_mpc_comm_test_mode), so that a one-GPU box still runs RCCL's ncclAllGather.
 * Exit code 0 = verified.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <hip/hip_runtime_api.h>

#include "tpc_mpc.h"

#define MAXG 16
#define CHECK_TPC(h, call)                                                                       \
    do {                                                                                         \
        const int rc_ = (call);                                                                  \
        if (rc_ != TPC_MPC_OK) {                                                                 \
            fprintf(stderr, "%s: status %d: %s\n", #call, rc_, tpc_mpc_last_error(h));           \
            return 10 + rc_;                                                                     \
        }                                                                                        \
    } while (0)
#define CHECK_HIP(call)                                                                          \
    do {                                                                                         \
        const hipError_t e_ = (call);                                                            \
        if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_)); return 5; } \
    } while (0)

int main(int argc, char** argv) {
    int devices = 0;
    if (hipGetDeviceCount(&devices) != hipSuccess || devices < 1) {
        fprintf(stderr, "no HIP device: there is no CPU path to fall back to\n");
        return 3;
    }
    int G = argc > 1 ? atoi(argv[1]) : devices;
    if (G < 1 || G > devices || G > MAXG) G = devices < MAXG ? devices : MAXG;
    const long n_total = argc > 2 ? atol(argv[2]) : 100003;   /* odd: ragged blocks for G > 1 */
    const int horizon = argc > 3 ? atoi(argv[3]) : 20;

    tpc_mpc_params p;
    if (tpc_mpc_default_params(&p, horizon) != TPC_MPC_OK) { fprintf(stderr, "unsupported horizon\n"); return 2; }
    /* AUTO picks a kernel family by batch size, and a block is smaller than the batch: pin the family, so that the
     * G-GPU result can be compared with the one-GPU result bit for bit (LANE is dlib's own arithmetic) */
    p.algo = TPC_MPC_ALGO_LANE;

    /* the whole batch on the host: a deterministic sweep over speeds and targets */
    double *v = malloc(sizeof(double) * n_total), *dy = malloc(sizeof(double) * n_total), *dphi = malloc(sizeof(double) * n_total);
    for (long k = 0; k < n_total; ++k) {
        v[k] = 0.1 + 3.9 * (double)(k % 977) / 976.0;
        dy[k] = -0.5 + (double)(k % 311) / 310.0;
        dphi[k] = -0.6 + 1.2 * (double)(k % 523) / 522.0;
    }

    tpc_mpc_handle h[MAXG];
    hipStream_t st[MAXG];
    double *d_v[MAXG], *d_dy[MAXG], *d_dphi[MAXG], *d_front[MAXG], *d_rear[MAXG];
    int64_t first[MAXG], count[MAXG];
    char id[TPC_MPC_COMM_ID_BYTES];
    for (int g = 0; g < G; ++g) {
        CHECK_TPC(NULL, tpc_mpc_create(g, &h[g]));
        CHECK_HIP(hipSetDevice(g));
        CHECK_HIP(hipStreamCreateWithFlags(&st[g], hipStreamNonBlocking));
        CHECK_TPC(NULL, tpc_mpc_shard_range(n_total, g, G, &first[g], &count[g]));
        CHECK_HIP(hipMalloc((void**)&d_v[g], sizeof(double) * (count[g] + 1)));
        CHECK_HIP(hipMalloc((void**)&d_dy[g], sizeof(double) * (count[g] + 1)));
        CHECK_HIP(hipMalloc((void**)&d_dphi[g], sizeof(double) * (count[g] + 1)));
        CHECK_HIP(hipMalloc((void**)&d_front[g], sizeof(double) * n_total));   /* FULL size: every GPU gets everything */
        CHECK_HIP(hipMalloc((void**)&d_rear[g], sizeof(double) * n_total));
        CHECK_HIP(hipMemcpy(d_v[g], v + first[g], sizeof(double) * count[g], hipMemcpyHostToDevice));
        CHECK_HIP(hipMemcpy(d_dy[g], dy + first[g], sizeof(double) * count[g], hipMemcpyHostToDevice));
        CHECK_HIP(hipMemcpy(d_dphi[g], dphi + first[g], sizeof(double) * count[g], hipMemcpyHostToDevice));
        CHECK_HIP(hipMemset(d_front[g], 0xff, sizeof(double) * n_total));
        CHECK_HIP(hipMemset(d_rear[g], 0xff, sizeof(double) * n_total));
        if (G == 1) CHECK_TPC(h[g], tpc_mpc_comm_test_mode(h[g], 1, 0));
    }
    /* one communicator; a single thread drives all ranks, so the inits are grouped */
    CHECK_TPC(NULL, tpc_mpc_comm_unique_id(id, sizeof id));
    CHECK_TPC(NULL, tpc_mpc_group_begin());
    for (int g = 0; g < G; ++g) CHECK_TPC(h[g], tpc_mpc_comm_init_rank(h[g], id, sizeof id, g, G));
    CHECK_TPC(NULL, tpc_mpc_group_end());

    /* one batch (twice: the second run reuses every allocation) */
    for (int rep = 0; rep < 2; ++rep) {
        CHECK_TPC(NULL, tpc_mpc_group_begin());
        for (int g = 0; g < G; ++g) {
            CHECK_HIP(hipSetDevice(g));
            CHECK_TPC(h[g], tpc_mpc_solve_batch_compact_sharded(h[g], &p, n_total, d_v[g], d_dy[g], d_dphi[g], d_front[g],
                                                                d_rear[g], NULL, NULL, st[g]));
        }
        CHECK_TPC(NULL, tpc_mpc_group_end());
        for (int g = 0; g < G; ++g) { CHECK_HIP(hipSetDevice(g)); CHECK_HIP(hipStreamSynchronize(st[g])); }
    }

    /* reference: GPU 0 solves the whole batch by itself (host arrays, a second handle) */
    double *want_f = malloc(sizeof(double) * n_total), *want_r = malloc(sizeof(double) * n_total);
    double *got_f = malloc(sizeof(double) * n_total), *got_r = malloc(sizeof(double) * n_total);
    tpc_mpc_handle whole;
    CHECK_TPC(NULL, tpc_mpc_create(0, &whole));
    CHECK_TPC(whole, tpc_mpc_solve_batch_compact(whole, &p, n_total, v, dy, dphi, want_f, want_r, NULL, NULL, TPC_MPC_HOST, NULL));
    long bad = 0;
    for (int g = 0; g < G; ++g) {
        CHECK_HIP(hipSetDevice(g));
        CHECK_HIP(hipMemcpy(got_f, d_front[g], sizeof(double) * n_total, hipMemcpyDeviceToHost));
        CHECK_HIP(hipMemcpy(got_r, d_rear[g], sizeof(double) * n_total, hipMemcpyDeviceToHost));
        if (memcmp(got_f, want_f, sizeof(double) * n_total) != 0 || memcmp(got_r, want_r, sizeof(double) * n_total) != 0) {
            for (long k = 0; k < n_total; ++k)
                if (memcmp(&got_f[k], &want_f[k], 8) != 0 || memcmp(&got_r[k], &want_r[k], 8) != 0) ++bad;
            fprintf(stderr, "GPU %d: %ld of %ld instances differ from the one-GPU solve\n", g, bad, n_total);
        }
        printf("{\"gpu\": %d, \"block\": [%lld, %lld], \"holds_all\": %s}\n", g, (long long)first[g],
               (long long)(first[g] + count[g]), bad == 0 ? "true" : "false");
    }
    printf("{\"gpus\": %d, \"n_total\": %ld, \"horizon\": %d, \"verified\": %s}\n", G, n_total, horizon, bad == 0 ? "true" : "false");
    for (int g = 0; g < G; ++g) {
        (void)hipSetDevice(g);
        tpc_mpc_destroy(h[g]);
        (void)hipStreamDestroy(st[g]);
        (void)hipFree(d_v[g]); (void)hipFree(d_dy[g]); (void)hipFree(d_dphi[g]); (void)hipFree(d_front[g]); (void)hipFree(d_rear[g]);
    }
    tpc_mpc_destroy(whole);
    free(v); free(dy); free(dphi); free(want_f); free(want_r); free(got_f); free(got_r);
    return bad == 0 ? 0 : 1;
}
