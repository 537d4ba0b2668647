/* examples/batch_compact.c -- plain-C host using the batch entry point with host memory.
 *
 *   gcc -std=c99 -Iinclude examples/batch_compact.c -Ltrajectory_controller_amd/lib -ltpc_mpc \
 *       -Wl,-rpath,$PWD/trajectory_controller_amd/lib -o batch_compact && ./batch_compact 20 1000
 *
 * Solves n synthetic (v, delta_y, delta_phi) instances -- what n calls of the reference's
 * mpcControllerTobi (src/trajectory_point_follower.cpp:301-389) would compute -- and prints the
 * first few steering pairs, the iteration statistics and the status flags.
 */
#include <stdio.h>
#include <stdlib.h>

#include "tpc_mpc.h"

int main(int argc, char** argv) {
    const int horizon = argc > 1 ? atoi(argv[1]) : 20;
    const long n = argc > 2 ? atol(argv[2]) : 1000;
    tpc_mpc_handle h = NULL;
    tpc_mpc_params p;
    if (tpc_mpc_default_params(&p, horizon) != TPC_MPC_OK) {
        fprintf(stderr, "unsupported horizon %d\n", horizon);
        return 2;
    }
    if (tpc_mpc_create(0, &h) != TPC_MPC_OK) {
        fprintf(stderr, "tpc_mpc_create: %s\n", tpc_mpc_last_error(NULL));
        return 3;   /* no GPU: there is no CPU path to fall back to */
    }
    double* v = malloc(sizeof(double) * n), *dy = malloc(sizeof(double) * n), *dphi = malloc(sizeof(double) * n);
    double* front = malloc(sizeof(double) * n), *rear = malloc(sizeof(double) * n);
    int32_t* iters = malloc(sizeof(int32_t) * n);
    for (long k = 0; k < n; ++k) {   /* a deterministic sweep over speeds and offsets */
        v[k] = 0.1 + 3.9 * (double)(k % 97) / 96.0;
        dy[k] = -0.5 + (double)(k % 31) / 30.0;
        dphi[k] = -0.6 + 1.2 * (double)(k % 53) / 52.0;
    }
    uint32_t flags = 0;
    const int rc = tpc_mpc_solve_batch_compact(h, &p, n, v, dy, dphi, front, rear, iters, &flags, TPC_MPC_HOST, NULL);
    if (rc != TPC_MPC_OK) {
        fprintf(stderr, "solve: %s\n", tpc_mpc_last_error(h));
        return 4;
    }
    long long total = 0;
    int max_it = 0;
    for (long k = 0; k < n; ++k) { total += iters[k]; if (iters[k] > max_it) max_it = iters[k]; }
    for (long k = 0; k < n && k < 5; ++k)
        printf("%ld v=%.17g dy=%.17g dphi=%.17g -> front=%.17g rear=%.17g iters=%d\n", k, v[k], dy[k], dphi[k],
               front[k], rear[k], iters[k]);
    printf("n=%ld horizon=%d mean_iters=%.1f max_iters=%d flags=0x%x\n", n, horizon, (double)total / (double)n, max_it, flags);
    free(v); free(dy); free(dphi); free(front); free(rear); free(iters);
    tpc_mpc_destroy(h);
    return 0;
}
