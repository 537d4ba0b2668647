// One translation unit per horizon (compile with -DTPC_LANE_H=<H>): instantiates the LANE kernels
// for fp64/fp32, compact and general (I = 1, 2) models, and exports their launchers.
#include <cstdint>

#include "mpc_lane.h"
#include "mpc_lanex.h"

#ifndef TPC_LANE_H
#error "compile with -DTPC_LANE_H=<horizon>"
#endif

namespace tpc {

namespace {

constexpr int kH = TPC_LANE_H;

// Persistent-wave count of a PG kernel: every wave must be resident at once (the queue is pulled,
// not pushed), so the grid is what the occupancy query says the chip holds, capped by the work.
// `block` threads per workgroup; at most 4 workgroups per CU are used (one per SIMD or SIMD pair).
// The answer depends on the device (CU count), so it is cached per device ordinal, once per kernel
// instantiation (`Tag` makes the cache unique to the call site).
template <class Tag, class Kernel>
inline int pg_grid(Kernel kernel, int block) {
    constexpr int kMaxDev = 64;
    static int cache[kMaxDev] = {0};   // 0 = not asked yet; racing first calls store the same value
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    if (dev >= 0 && dev < kMaxDev && cache[dev] > 0) return cache[dev];
    int cus = 256, per_cu = 4;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, block, 0) != hipSuccess || per_cu < 1)
        per_cu = 1;
    if (per_cu > 4) per_cu = 4;
    if (dev >= 0 && dev < kMaxDev) cache[dev] = cus * per_cu;
    return cus * per_cu;
}
struct TagState {}; struct TagFast {}; struct TagExact {}; struct TagLanex {}; template <int I> struct TagLanexG {};

// The fused PG kernel publishes controls[0] only; a caller that wants the controller state back
// (warm-start chains, tpc_mpc_rollout) gets the kernel that keeps it.
inline bool wants_state(const CompactArgs&) { return false; }
inline bool wants_state(const GeneralArgs& a) { return a.controls != nullptr || a.v != nullptr; }

// coordinate descent + queue order (also the front half of the general-form GROUP kernels where the caller passes the
// controller state or the horizon has no LANE_FMA kernel: mpc_groupg_inst.hip)
template <typename T, int I, class Model, class Args>
hipError_t phase1(const Args& a, const Knobs& k, const Workspace& ws, hipStream_t s) {
    T* recs = (T*)ws.state;
    hipError_t e = hipMemsetAsync(ws.ticket, 0, sizeof(uint32_t), s);
    if (e == hipSuccess) e = hipMemsetAsync(ws.stats, 0, 3 * sizeof(unsigned long long), s);
    if (e == hipSuccess) e = order_begin(ws.sort_temp, s);
    if (e != hipSuccess) return e;
    const int cd_grid = (int)((a.n + kWave - 1) / kWave);
    if (ws.ev) (void)hipEventRecord(ws.ev[0], s);
    hipLaunchKernelGGL((lane_cd_kernel<T, I, kH, Model, Args>), dim3(cd_grid), dim3(kWave), 0, s, a, k, recs,
                       ws.keys, ws.rank, (uint32_t*)ws.sort_temp, ws.stats, wants_state(a) ? 0 : 1);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    e = order_finish(ws.keys, ws.rank, ws.order, a.n, ws.sort_temp, s);
    if (e != hipSuccess) return e;
    if (ws.ev) (void)hipEventRecord(ws.ev[1], s);
    return hipSuccess;
}
// the projected-gradient launches; `only_if_refused`: behind a GROUP kernel, i.e. only the build(s) that serve a batch
// the stop-test screen refused
template <typename T, int I, class Model, class Args>
hipError_t phase2(const Args& a, const Knobs& k, const Workspace& ws, hipStream_t s, bool only_if_refused) {
    T* recs = (T*)ws.state;
    if (wants_state(a)) {
        const int grid_cap = pg_grid<TagState>(lane_pg_kernel<T, I, kH, Model, Args>, kWave);
        const int64_t need = (a.n + kWave - 1) / kWave;
        hipLaunchKernelGGL((lane_pg_kernel<T, I, kH, Model, Args>), dim3((unsigned)(need < grid_cap ? need : grid_cap)),
                           dim3(kWave), 0, s, a, k, (const T*)recs, (const uint32_t*)ws.order, ws.ticket, ws.stats,
                           only_if_refused ? 1 : 0);
    } else {
        constexpr int bt = kWave * FusedOcc<T, kH>::value;
        const int64_t need = (a.n + bt - 1) / bt;
        // first queue position of the instances the CD kernel published itself (order_finish)
        const uint32_t* queue_len = order_queue_len(ws.sort_temp);
        if constexpr (Model::kFastStop) {
            // both builds go out; the one the CD kernel's screen did not pick returns at once
            if (!only_if_refused) {
                const int fast_cap = pg_grid<TagFast>(lane_pg_fused_kernel<T, I, kH, Model, Args, true>, bt);
                hipLaunchKernelGGL((lane_pg_fused_kernel<T, I, kH, Model, Args, true>),
                                   dim3((unsigned)(need < fast_cap ? need : fast_cap)), dim3(bt), 0, s, a, k,
                                   (const T*)recs, (const uint32_t*)ws.order, ws.ticket, ws.stats, queue_len);
            }
        }
        const int grid_cap = pg_grid<TagExact>(lane_pg_fused_kernel<T, I, kH, Model, Args, false>, bt);
        hipLaunchKernelGGL((lane_pg_fused_kernel<T, I, kH, Model, Args, false>),
                           dim3((unsigned)(need < grid_cap ? need : grid_cap)), dim3(bt), 0, s, a, k,
                           (const T*)recs, (const uint32_t*)ws.order, ws.ticket, ws.stats, queue_len);
    }
    return hipGetLastError();
}

// Batch sizes below which the compact fp64 LANE family runs its projected-gradient phase G lanes per instance
// (mpc_lanex.h), measured on a 256-CU part (profiles/r04_lanex_crossover.txt) and scaled by the CU count.
inline int64_t lanex_below(const Workspace& ws, bool general) {
    if (ws.lanex_below >= 0) return ws.lanex_below;
    // (compact N = 40: 14.3 against 49.8 ms up to 16 384, level at 65 536; general form, two inputs: 21.6 against 61.4,
    // level near 60 000 -- one input near 45 000; profiles/r04_lanex_crossover*.txt)
    const int64_t at = general ? (kH == 40 ? 49152 : (kH == 30 ? 57344 : (kH == 20 ? 81920 : (kH == 10 ? 98304 : 0))))
                               : (kH == 40 ? 60000 : (kH == 30 ? 65536 : (kH == 20 || kH == 10 ? 98304 : 0)));
    const int cus = ws.cu_count > 0 ? ws.cu_count : 256;
    return at * cus / 256 < at ? at * cus / 256 : at;
}

template <typename T, int I, class Model, class Args>
hipError_t run(const Args& a, const Knobs& k, const Workspace& ws, hipStream_t s) {
    if (a.n <= 0) return hipSuccess;
    hipError_t e = phase1<T, I, Model, Args>(a, k, ws, s);
    if (e != hipSuccess) return e;
    if constexpr (std::is_same<Model, CompactModel<T>>::value && sizeof(T) == 8 && LanexPlan<kH>::built) {
        if (a.n < lanex_below(ws, false)) {
            constexpr int ng = LanexPlan<kH>::NG;
            const int64_t need = (a.n + ng - 1) / ng;
            const int grid_cap = pg_grid<TagLanex>(lanex_pg_kernel<T, kH>, kWave);
            hipLaunchKernelGGL((lanex_pg_kernel<T, kH>), dim3((unsigned)(need < grid_cap ? need : grid_cap)), dim3(kWave), 0, s, a, k,
                               (const T*)ws.state, (const uint32_t*)ws.order, ws.ticket, ws.stats, order_queue_len(ws.sort_temp),
                               GroupRefillBatch<LanexPlan<kH>::G>::value);
            e = hipGetLastError();
            if (ws.ev) (void)hipEventRecord(ws.ev[2], s);
            return e;
        }
    }
    if constexpr (std::is_same<Args, GeneralArgs>::value && sizeof(T) == 8 && LanexPlan<kH>::built) {
        if (a.n < lanex_below(ws, true)) {   // the same kernel family for the general model
            constexpr int ng = LanexPlan<kH>::NG;
            const int64_t need = (a.n + ng - 1) / ng;
            if (wants_state(a)) {            // controller state in / out: every instance through the kernel (lane_pg_kernel's job)
                const int grid_cap = pg_grid<TagLanexG<I + 2>>(lanexg_pg_kernel<T, I, kH, true>, kWave);
                hipLaunchKernelGGL((lanexg_pg_kernel<T, I, kH, true>), dim3((unsigned)(need < grid_cap ? need : grid_cap)), dim3(kWave), 0, s,
                                   a, k, (const T*)ws.state, (const uint32_t*)ws.order, ws.ticket, ws.stats, (const uint32_t*)nullptr,
                                   GroupRefillBatch<LanexPlan<kH>::G>::value);
            } else {
                const int grid_cap = pg_grid<TagLanexG<I>>(lanexg_pg_kernel<T, I, kH>, kWave);
                hipLaunchKernelGGL((lanexg_pg_kernel<T, I, kH>), dim3((unsigned)(need < grid_cap ? need : grid_cap)), dim3(kWave), 0, s, a, k,
                                   (const T*)ws.state, (const uint32_t*)ws.order, ws.ticket, ws.stats, order_queue_len(ws.sort_temp),
                                   GroupRefillBatch<LanexPlan<kH>::G>::value);
            }
            e = hipGetLastError();
            if (ws.ev) (void)hipEventRecord(ws.ev[2], s);
            return e;
        }
    }
    e = phase2<T, I, Model, Args>(a, k, ws, s, false);
    if (ws.ev) (void)hipEventRecord(ws.ev[2], s);
    return e;
}

// The instances of a batch that ended on the iteration cap, once more in dlib's own arithmetic: lane_cd_kernel
// (RESOLVE) leaves their records and queues them, the fused projected-gradient kernel finishes them.  `gate` closed
// (no instance on the cap): one 32-byte memset and two kernels that leave at once.
//
// SUBSET 2 (presolve): the same two kernels BEFORE the tolerance family's pass, on a stream of their own, for the instances
// lambda predicts to end on the cap (lane_cd_kernel); SUBSET 1 then leaves those out.
template <typename T, int I, class Model, class Args, int SUBSET = 1>
hipError_t resolve(const Args& a, const Knobs& k, const Workspace& ws, const int32_t* select, const uint32_t* gate, hipStream_t s,
                   double lambda_from = 0.0, const uint32_t* pre_len = nullptr, uint32_t pre_limit = 0u) {
    if (a.n <= 0) return hipSuccess;
    T* recs = (T*)ws.state;
    // ticket, queue length and statistics of this stage sit side by side (resolve_workspace, tpc_mpc_api.cpp): one memset
    uint32_t* queue_len = ws.ticket + 1;
    hipError_t e = hipMemsetAsync(ws.ticket, 0, 32, s);
    if (e != hipSuccess) return e;
    const int grid = (int)((a.n + kWave - 1) / kWave);
    hipLaunchKernelGGL((lane_cd_kernel<T, I, kH, Model, Args, SUBSET>), dim3(grid), dim3(kWave), 0, s, a, k, recs, ws.order,
                       (uint32_t*)nullptr, queue_len, ws.stats, 1, select, gate, (T)lambda_from, pre_len, pre_limit);
    if constexpr (std::is_same<Model, CompactModel<T>>::value && sizeof(T) == 8 && LanexPlan<kH>::built) {
        // compact form, N = 10, 20, 40: G lanes per instance, the chains handed from chunk to chunk (mpc_lanex.h) -- the
        // same bits at a third of the iteration's time
        constexpr int ng = LanexPlan<kH>::NG;
        const int64_t need = (a.n + ng - 1) / ng;
        if constexpr (SUBSET == 2) {   // beside the tolerance family's grid: SIMDs of its own, one round of the chain or nothing
            const int64_t solo = (int64_t)pre_limit / ng;
            hipLaunchKernelGGL((lanex_pg_kernel<T, kH, true>), dim3((unsigned)(need < solo ? need : solo)), dim3(kWave), 0, s, a, k,
                               (const T*)recs, (const uint32_t*)ws.order, ws.ticket, ws.stats, (const uint32_t*)queue_len, 1, pre_limit);
            return hipGetLastError();
        }
        const int grid_cap = pg_grid<TagLanex>(lanex_pg_kernel<T, kH>, kWave);
        hipLaunchKernelGGL((lanex_pg_kernel<T, kH>), dim3((unsigned)(need < grid_cap ? need : grid_cap)), dim3(kWave), 0, s, a, k,
                           (const T*)recs, (const uint32_t*)ws.order, ws.ticket, ws.stats, (const uint32_t*)queue_len, 1);
        return hipGetLastError();
    } else if constexpr (std::is_same<Args, GeneralArgs>::value && sizeof(T) == 8 && LanexPlan<kH>::built) {
        constexpr int ng = LanexPlan<kH>::NG;
        const int64_t need = (a.n + ng - 1) / ng;
        if constexpr (SUBSET == 2) {
            const int64_t solo = (int64_t)pre_limit / ng;
            hipLaunchKernelGGL((lanexg_pg_kernel<T, I, kH, false, true>), dim3((unsigned)(need < solo ? need : solo)), dim3(kWave), 0, s,
                               a, k, (const T*)recs, (const uint32_t*)ws.order, ws.ticket, ws.stats, (const uint32_t*)queue_len, 1, pre_limit);
            return hipGetLastError();
        }
        const int grid_cap = pg_grid<TagLanexG<I>>(lanexg_pg_kernel<T, I, kH>, kWave);
        hipLaunchKernelGGL((lanexg_pg_kernel<T, I, kH>), dim3((unsigned)(need < grid_cap ? need : grid_cap)), dim3(kWave), 0, s, a, k,
                           (const T*)recs, (const uint32_t*)ws.order, ws.ticket, ws.stats, (const uint32_t*)queue_len, 1);
        return hipGetLastError();
    } else {
    constexpr int bt = kWave * FusedOcc<T, kH>::value;
    const int64_t need = (a.n + bt - 1) / bt;
    // the exact-stop-test build only (bit-exact like the other one; lane_cd_kernel raises stats[2] for it where it
    // queued anything): one launch fewer on the many calls that have nothing to re-solve
    const int grid_cap = pg_grid<TagExact>(lane_pg_fused_kernel<T, I, kH, Model, Args, false>, bt);
    hipLaunchKernelGGL((lane_pg_fused_kernel<T, I, kH, Model, Args, false>), dim3((unsigned)(need < grid_cap ? need : grid_cap)),
                       dim3(bt), 0, s, a, k, (const T*)recs, (const uint32_t*)ws.order, ws.ticket, ws.stats, (const uint32_t*)queue_len);
    return hipGetLastError();
    }
}

}  // namespace

#define TPC_CAT2(a, b) a##b
#define TPC_CAT(a, b) TPC_CAT2(a, b)

// Which LLVM machine scheduler built this translation unit (csrc/Makefile passes it): readable
// through tpc_mpc_build_info(), so a shipped object says how it was made.
#ifndef TPC_SCHED_NAME
#define TPC_SCHED_NAME "default"
#endif
const char* TPC_CAT(lane_build_h, TPC_LANE_H)() { return "sched=" TPC_SCHED_NAME; }

int64_t TPC_CAT(lane_rec_len_h, TPC_LANE_H)(int dtype) {
    return dtype == 0 ? LaneRec<double, kH>::kLen : LaneRec<float, kH>::kLen;
}

hipError_t TPC_CAT(lane_compact_h, TPC_LANE_H)(int dtype, const CompactArgs& a, const Knobs& k,
                                                 const Workspace& ws, hipStream_t s) {
    if (dtype == 0) return run<double, 2, CompactModel<double>, CompactArgs>(a, k, ws, s);
    return run<float, 2, CompactModel<float>, CompactArgs>(a, k, ws, s);
}

// fp64 only: the re-solve exists to deliver dlib's bits, and dlib is fp64
hipError_t TPC_CAT(lane_resolve_compact_h, TPC_LANE_H)(const CompactArgs& a, const Knobs& k, const Workspace& ws,
                                                         const int32_t* select, const uint32_t* gate, hipStream_t s,
                                                         double presolved_from, const uint32_t* pre_len, uint32_t pre_limit) {
    return resolve<double, 2, CompactModel<double>, CompactArgs>(a, k, ws, select, gate, s, presolved_from, pre_len, pre_limit);
}
// the instances with lambda >= lambda_from, ahead of the tolerance family (outputs where `a` says: the caller's side arrays)
hipError_t TPC_CAT(lane_presolve_compact_h, TPC_LANE_H)(const CompactArgs& a, const Knobs& k, const Workspace& ws,
                                                          double lambda_from, uint32_t limit, hipStream_t s) {
    if constexpr (!LanexPlan<kH>::built) return hipErrorInvalidValue;
    return resolve<double, 2, CompactModel<double>, CompactArgs, 2>(a, k, ws, nullptr, nullptr, s, lambda_from, nullptr, limit);
}
hipError_t TPC_CAT(lane_resolve_general_h, TPC_LANE_H)(int I, const GeneralArgs& a, const Knobs& k, const Workspace& ws,
                                                         const int32_t* select, const uint32_t* gate, hipStream_t s,
                                                         double presolved_from, const uint32_t* pre_len, uint32_t pre_limit) {
    if (I == 2) return resolve<double, 2, GeneralModel<double, 2>, GeneralArgs>(a, k, ws, select, gate, s, presolved_from, pre_len, pre_limit);
    return resolve<double, 1, GeneralModel<double, 1>, GeneralArgs>(a, k, ws, select, gate, s, presolved_from, pre_len, pre_limit);
}
hipError_t TPC_CAT(lane_presolve_general_h, TPC_LANE_H)(int I, const GeneralArgs& a, const Knobs& k, const Workspace& ws,
                                                          double lambda_from, uint32_t limit, hipStream_t s) {
    if constexpr (!LanexPlan<kH>::built) return hipErrorInvalidValue;
    if (I == 2) return resolve<double, 2, GeneralModel<double, 2>, GeneralArgs, 2>(a, k, ws, nullptr, nullptr, s, lambda_from, nullptr, limit);
    return resolve<double, 1, GeneralModel<double, 1>, GeneralArgs, 2>(a, k, ws, nullptr, nullptr, s, lambda_from, nullptr, limit);
}

// the halves the general-form GROUP kernels borrow (fp64; mpc_groupg_inst.hip): coordinate descent + queue order, and
// the bit-exact projected-gradient kernels for a batch the stop-test screen refused
hipError_t TPC_CAT(lane_general_phase1_h, TPC_LANE_H)(int I, const GeneralArgs& a, const Knobs& k, const Workspace& ws, hipStream_t s) {
    if (I == 2) return phase1<double, 2, GeneralModel<double, 2>, GeneralArgs>(a, k, ws, s);
    return phase1<double, 1, GeneralModel<double, 1>, GeneralArgs>(a, k, ws, s);
}
hipError_t TPC_CAT(lane_general_refused_h, TPC_LANE_H)(int I, const GeneralArgs& a, const Knobs& k, const Workspace& ws, hipStream_t s) {
    if (I == 2) return phase2<double, 2, GeneralModel<double, 2>, GeneralArgs>(a, k, ws, s, true);
    return phase2<double, 1, GeneralModel<double, 1>, GeneralArgs>(a, k, ws, s, true);
}

hipError_t TPC_CAT(lane_general_h, TPC_LANE_H)(int dtype, int I, const GeneralArgs& a, const Knobs& k,
                                                 const Workspace& ws, hipStream_t s) {
    if (dtype == 0) {
        if (I == 2) return run<double, 2, GeneralModel<double, 2>, GeneralArgs>(a, k, ws, s);
        return run<double, 1, GeneralModel<double, 1>, GeneralArgs>(a, k, ws, s);
    }
    if (I == 2) return run<float, 2, GeneralModel<float, 2>, GeneralArgs>(a, k, ws, s);
    return run<float, 1, GeneralModel<float, 1>, GeneralArgs>(a, k, ws, s);
}

}  // namespace tpc
