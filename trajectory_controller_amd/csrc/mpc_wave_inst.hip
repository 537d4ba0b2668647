// One translation unit per horizon (compile with -DTPC_WAVE_H=<H>): instantiates the WAVE kernel
// for fp64/fp32, compact and general (I = 1, 2) models, and exports its launchers.  Horizons with
// I*H > 64 have no WAVE kernel; their launchers report hipErrorNotSupported.
#include <cstdlib>

#include "mpc_wave.h"

#ifndef TPC_WAVE_H
#error "compile with -DTPC_WAVE_H=<horizon>"
#endif

namespace tpc {

namespace {

constexpr int kH = TPC_WAVE_H;

// compute units of the current device, cached per device ordinal
int cu_count() {
    constexpr int kMaxDev = 64;
    static int cache[kMaxDev] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDev) return 256;
    if (cache[dev] <= 0) {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        cache[dev] = cus;
    }
    return cache[dev];
}

template <typename T, int I, class Model, class Args>
hipError_t run(const Args& a, const Knobs& k, const Workspace& ws, hipStream_t s) {
    if constexpr (I * kH > 2 * kWave || (I * kH > kWave && I != 2)) {
        return hipErrorNotSupported;
    } else {
        if (a.n <= 0) return hipSuccess;
        if (a.n > 0x7fffffffll) return hipErrorInvalidValue;
        if (ws.ev) (void)hipEventRecord(ws.ev[0], s);
        // more instances than the persistent grid holds: wavefronts over a longest-first queue (mpc_wave.h)
        constexpr int wpb = waves_per_block<T, I, kH, Model>();
        const int64_t slots = (int64_t)cu_count() * queue_waves_per_cu<I, kH, T, Model>();
        // (not for the general form with two variables per lane: its fp64 queue kernel -- the model's registers on
        // top of two Hessian rows -- spills into scratch inside the loops; the plain launch below does not)
        constexpr bool queue_ok = kH >= kQueueMinHorizon &&
                                  !(wave_two_per_lane<I, kH>() && !std::is_same<Model, CompactModel<T>>::value);
        bool queued = false;
        // fp64, at most 32 variables: two instances per wavefront over the same queue (mpc_wave.h, wave_pair_solve)
        // (at every horizon: paired, even N = 4 and 5 are long enough for the queue to pay -- 280 -> 239 us for
        // 32 768 instances at N = 5, 207 -> 195 at N = 4)
        constexpr bool pair_ok = sizeof(T) == 8 && I * kH <= kWave / 2;
        // TPC_MPC_OPT_WAVE_GROUP = 1 keeps strictly one instance per wavefront (for A/B measurements)
        const bool pairs_wanted = ws.wave_group != 1;
        if constexpr (pair_ok) {
            // from more than one wavefront per SIMD on: 2 048 instances at N = 10 take 141 us one per wavefront (two
            // wavefronts sharing every SIMD) and ~120 us as 1 024 pairs of neighbours in the queue order
            if (pairs_wanted && a.n > slots / kQueueWorkgroupsPerCu && a.n <= kQueueMaxInstances && ws.order) {
                hipLaunchKernelGGL((wave_order_kernel<T, I, kH, Model, Args>), dim3(1), dim3(kOrderThreads), 0, s, a,
                                   ws.order, ws.ticket);
                // at most 16 variables: four instances per wavefront, one per 16-lane row, once pairs would put more
                // than one wavefront on a SIMD (32 768 x N = 4: 198 us in pairs, 147 in fours; 4 096: 47 / 44;
                // TPC_MPC_OPT_WAVE_GROUP = 2 | 4 forces either)
                constexpr bool quad_ok = I * kH <= kWave / 4;
                const int group_forced = ws.wave_group;
                bool quads = false;
                if constexpr (quad_ok) quads = group_forced == 4 || (group_forced == 0 && a.n > slots);
                if constexpr (quad_ok) {
                    if (quads)
                        hipLaunchKernelGGL((wave_pair_queue_kernel<T, I, kH, 4, Model, Args>), dim3((unsigned)(slots / kWavesPerBlock)),
                                           dim3(kWavesPerBlock * kWave), 0, s, a, k, (const uint32_t*)ws.order, ws.ticket);
                }
                if (!quads)
                    hipLaunchKernelGGL((wave_pair_queue_kernel<T, I, kH, 2, Model, Args>), dim3((unsigned)(slots / kWavesPerBlock)),
                                       dim3(kWavesPerBlock * kWave), 0, s, a, k, (const uint32_t*)ws.order, ws.ticket);
                queued = true;
            }
        }
        if constexpr (queue_ok) {
            if (!queued && a.n > slots && a.n <= kQueueMaxInstances && ws.order) {
                hipLaunchKernelGGL((wave_order_kernel<T, I, kH, Model, Args>), dim3(1), dim3(kOrderThreads), 0, s, a,
                                   ws.order, ws.ticket);
                hipLaunchKernelGGL((wave_queue_kernel<T, I, kH, Model, Args>), dim3((unsigned)(slots / wpb)),
                                   dim3(wpb * kWave), 0, s, a, k, (const uint32_t*)ws.order, ws.ticket);
                queued = true;
            }
        }
        if (!queued)
            hipLaunchKernelGGL((wave_kernel<T, I, kH, Model, Args>), dim3((unsigned)((a.n + wpb - 1) / wpb)),
                               dim3(wpb * kWave), 0, s, a, k);
        const hipError_t e = hipGetLastError();
        if (ws.ev) { (void)hipEventRecord(ws.ev[1], s); (void)hipEventRecord(ws.ev[2], s); }
        return e;
    }
}

}  // namespace

#define TPC_CAT2(a, b) a##b
#define TPC_CAT(a, b) TPC_CAT2(a, b)

hipError_t TPC_CAT(wave_compact_h, TPC_WAVE_H)(int dtype, const CompactArgs& a, const Knobs& k, const Workspace& ws, hipStream_t s) {
    if (dtype == 0) return run<double, 2, CompactModel<double>, CompactArgs>(a, k, ws, s);
    return run<float, 2, CompactModel<float>, CompactArgs>(a, k, ws, s);
}

hipError_t TPC_CAT(wave_general_h, TPC_WAVE_H)(int dtype, int I, const GeneralArgs& a, const Knobs& k, const Workspace& ws, hipStream_t s) {
    if (dtype == 0) {
        if (I == 2) return run<double, 2, GeneralModel<double, 2>, GeneralArgs>(a, k, ws, s);
        return run<double, 1, GeneralModel<double, 1>, GeneralArgs>(a, k, ws, s);
    }
    if (I == 2) return run<float, 2, GeneralModel<float, 2>, GeneralArgs>(a, k, ws, s);
    return run<float, 1, GeneralModel<float, 1>, GeneralArgs>(a, k, ws, s);
}

}  // namespace tpc
