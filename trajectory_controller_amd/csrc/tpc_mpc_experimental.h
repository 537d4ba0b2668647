/* Experimental entry points of libtpc_mpc.so that are NOT part of the drop-in boundary (include/tpc_mpc.h):
 * exported for this repository's own measurements and tests, may change or go. */
#ifndef TPC_MPC_EXPERIMENTAL_H
#define TPC_MPC_EXPERIMENTAL_H

#include "../../include/tpc_mpc.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Queue-order hint.  Instances of a batch need between a few and several thousand iterations (mpc.h:271,
 * :310) and a batch finishes when its slowest lane does, so the LANE families -- and the WAVE kernels over
 * their work queue -- start the instances expected to run longest first; their own estimate is dlib's lambda
 * (mpc.h:116-123).  hint[k] = expected iteration count of instance k replaces that estimate for the NEXT
 * compact / general batch solve of this handle with the same n (copied into the handle by this call, forgotten
 * after one solve; NULL clears).  It never changes a result, only which lane solves which instance when.
 * Measured (scripts/hint_gain.py): exact counts of the same instances buy 7 % at N = 20; counts one control
 * cycle old, inputs moved by 0.5 % of their range, buy nothing -- which is why this is not in the public header. */
int tpc_mpc_x_set_work_hint(tpc_mpc_handle h, const int32_t* hint, int64_t n, int mem);

/* The share of the chip a handle's GROUP solves take: `waves` persistent wavefronts (0 = one per SIMD, the default) and
 * the CU count AUTO's crossover table is scaled to (0 = back to the device's own).  What tpc_mpc_solve_batch_compact_mixed sets on
 * the child handles of its bins; exported to measure one bin on a share by itself (scripts/group_share.py). */
int tpc_mpc_x_set_group_share(tpc_mpc_handle h, int waves, int cu_count);

/* The bit-exact LANE family, compact form, fp64, N = 10 / 20 / 30 / 40: batches of fewer than `below` instances run their
 * projected-gradient phase G lanes per instance (csrc/mpc_lanex.h: same bits, a third of the iteration's latency, a third
 * of the throughput of a full chip).  -1 (default): the measured crossover; 0: never.  Exported to measure that crossover
 * (scripts/lanex_crossover.py). */
int tpc_mpc_x_set_lanex_below(tpc_mpc_handle h, int64_t below);

/* The collectives tpc_mpc_solve_batch_compact_sharded_split / tpc_mpc_gather_shards_split issue for ONE output row, as data:
 * ops5[5 * i + 0..4] = kind (0 all-gather: every rank sends `count` elements at send_off and receives world * count at
 * recv_off; 1 in-place broadcast of [send_off, +count) from `root`), root, send_off, recv_off, count -- offsets in elements
 * of the exchanged buffer, which is the full-size row (block split) or the [world][ceil(n / world)] staging array
 * (interleaved split; *buffer_elems says how long).  Returns the number of operations (-1: bad arguments).  No device, no
 * RCCL: it lets the CPU suite execute the slot arithmetic of all `world` owners with a host-memory communicator. */
int tpc_mpc_x_exchange_plan(int64_t n_total, int world, int rank, int split, int force_ragged, int64_t* ops5, int max_ops,
                            int64_t* buffer_elems);

#ifdef __cplusplus
}
#endif
#endif
