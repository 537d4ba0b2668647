// One translation unit per horizon (compile with -DTPC_WAVE_H=<H>): instantiates the WAVE kernel
// for fp64/fp32, compact and general (I = 1, 2) models, and exports its launchers.  Horizons with
// I*H > 64 have no WAVE kernel; their launchers report hipErrorNotSupported.
#include "mpc_wave.h"

#ifndef TPC_WAVE_H
#error "compile with -DTPC_WAVE_H=<horizon>"
#endif

namespace tpc {

namespace {

constexpr int kH = TPC_WAVE_H;

template <typename T, int I, class Model, class Args>
hipError_t run(const Args& a, const Knobs& k, const Workspace& ws, hipStream_t s) {
    if constexpr (I * kH > kWave) {
        return hipErrorNotSupported;
    } else {
        if (a.n <= 0) return hipSuccess;
        if (a.n > 0x7fffffffll) return hipErrorInvalidValue;
        if (ws.ev) (void)hipEventRecord(ws.ev[0], s);
        hipLaunchKernelGGL((wave_kernel<T, I, kH, Model, Args>), dim3((unsigned)((a.n + kWavesPerBlock - 1) / kWavesPerBlock)),
                           dim3(kWavesPerBlock * kWave), 0, s, a, k);
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
