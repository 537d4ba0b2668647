/* Latency of tpc_mpc_solve_one -- the reference's real call pattern: ONE horizon-N solve per cycle()
 * (reference: src/trajectory_point_follower.cpp:366-380; MPC_HORIZON = 4, include/...follower.h:48).
 * Plain C99 against include/tpc_mpc.h:
 *     gcc -std=c99 -O2 -Iinclude examples/solve_one_latency.c -Ltrajectory_controller_amd/lib -ltpc_mpc
 * Prints, per horizon: the resident path (default) with a new speed in every call and with the speed held (the
 * wave then reuses the model-only part of its set-up), the same with max_iter = 0 (mailbox round trip + set-up only,
 * no iterations) and the launch path (tpc_mpc_set_resident(h, 0)).
 *     ./a.out [reps] host      the host path only: a handle created with TPC_MPC_DEVICE_NONE (no GPU needed) */
#define _POSIX_C_SOURCE 199309L
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "tpc_mpc.h"

static double now_us(void) {
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return t.tv_sec * 1e6 + t.tv_nsec * 1e-3;
}

static int cmp(const void* a, const void* b) { return (*(const double*)a > *(const double*)b) - (*(const double*)a < *(const double*)b); }

/* vary_v = 0: the speed stays (only the targets change from call to call), so the resident wave reuses the part of
 * the set-up it derives from the model alone; 1: a new speed in every call recomputes it. */
static void run(tpc_mpc_handle h, const tpc_mpc_params* p, int reps, const char* what, int vary_v) {
    double* t = (double*)malloc(sizeof(double) * reps);
    double f = 0, r = 0, acc = 0;
    for (int i = 0; i < 50; ++i) tpc_mpc_solve_one(h, p, 1.0, 0.1, 0.05, &f, &r);
    for (int i = 0; i < reps; ++i) {
        const double v = vary_v ? 0.5 + 3.0 * ((i * 2654435761u) % 1000) / 1000.0 : 1.7;
        const double t0 = now_us();
        if (tpc_mpc_solve_one(h, p, v, 0.1 + 1e-4 * (i % 97), -0.2 + 4e-3 * (i % 89), &f, &r) != TPC_MPC_OK) {
            fprintf(stderr, "solve_one: %s\n", tpc_mpc_last_error(h));
            exit(2);
        }
        t[i] = now_us() - t0;
        acc += f + r;
    }
    qsort(t, reps, sizeof(double), cmp);
    double mean = 0;
    for (int i = 0; i < reps; ++i) mean += t[i] / reps;
    printf("N=%2d %-30s mean %7.2f us  p50 %7.2f  p90 %7.2f  p99 %7.2f  min %7.2f   (chk %.6f)\n", p->horizon, what, mean,
           t[reps / 2], t[reps * 9 / 10], t[reps * 99 / 100], t[0], acc);
    free(t);
}

int main(int argc, char** argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 2000;
    tpc_mpc_handle h;
    if (argc > 2 && strcmp(argv[2], "host") == 0) {   /* the calling thread solves: csrc/tpc_mpc_host.cpp */
        const int hh[] = {4, 5, 10, 20};
        if (tpc_mpc_create(TPC_MPC_DEVICE_NONE, &h) != TPC_MPC_OK) { fprintf(stderr, "tpc_mpc_create: %s\n", tpc_mpc_last_error(NULL)); return 3; }
        for (int k = 0; k < 4; ++k) {
            tpc_mpc_params p;
            tpc_mpc_default_params(&p, hh[k]);
            run(h, &p, reps, "host path, new v per call", 1);
        }
        tpc_mpc_destroy(h);
        return 0;
    }
    if (tpc_mpc_create(0, &h) != TPC_MPC_OK) { fprintf(stderr, "tpc_mpc_create: %s\n", tpc_mpc_last_error(NULL)); return 3; }
    const int hs[] = {4, 10, 20, 30, 40};
    for (int k = 0; k < 5; ++k) {
        const int reps_all = reps;
        const int reps = hs[k] >= 30 ? (reps_all + 9) / 10 : reps_all;   /* milliseconds per call up there */
        tpc_mpc_params p;
        tpc_mpc_default_params(&p, hs[k]);
        tpc_mpc_set_resident(h, 20000);
        run(h, &p, reps, "resident, new v per call", 1);
        run(h, &p, reps, "resident, same v", 0);
        p.max_iter = 0;
        run(h, &p, reps, "resident, max_iter=0", 1);
        run(h, &p, reps, "resident, max_iter=0, same v", 0);
        p.max_iter = 10000;
        tpc_mpc_set_resident(h, 0);
        run(h, &p, reps, "launch per call", 1);
        p.max_iter = 0;
        run(h, &p, reps, "launch per call, max_iter=0", 1);
    }
    tpc_mpc_destroy(h);
    return 0;
}
