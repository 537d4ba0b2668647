// C ABI of libtpc_mpc.so (include/tpc_mpc.h): argument validation, host<->device staging, kernel
// family selection.  No solver arithmetic lives here and there is no CPU solve path: every entry
// point ends in a gfx950 kernel launch or fails.  (Mixed-horizon batches: tpc_mpc_mixed.hip; sharded
// solves over RCCL: tpc_mpc_comm.cpp; the resident single-solve kernel: tpc_mpc_one.hip.)
#include "tpc_mpc_context.h"
#include "tpc_mpc_experimental.h"
#include "auto_table.h"

#include <cmath>
#include <cstdlib>
#include <mutex>

namespace tpc {
// per-horizon launchers, one translation unit each (mpc_lane_inst.hip / mpc_wave_inst.hip)
#define TPC_DECL_H(h)                                                                             \
    hipError_t lane_compact_h##h(int, const CompactArgs&, const Knobs&, const Workspace&, hipStream_t); \
    hipError_t lane_general_h##h(int, int, const GeneralArgs&, const Knobs&, const Workspace&, hipStream_t); \
    hipError_t lane_resolve_compact_h##h(const CompactArgs&, const Knobs&, const Workspace&, const int32_t*, const uint32_t*, hipStream_t, double, const uint32_t*, uint32_t); \
    hipError_t lane_presolve_compact_h##h(const CompactArgs&, const Knobs&, const Workspace&, double, uint32_t, hipStream_t); \
    hipError_t lane_resolve_general_h##h(int, const GeneralArgs&, const Knobs&, const Workspace&, const int32_t*, const uint32_t*, hipStream_t, double, const uint32_t*, uint32_t); \
    hipError_t lane_presolve_general_h##h(int, const GeneralArgs&, const Knobs&, const Workspace&, double, uint32_t, hipStream_t); \
    hipError_t wave_compact_h##h(int, const CompactArgs&, const Knobs&, const Workspace&, hipStream_t); \
    hipError_t wave_general_h##h(int, int, const GeneralArgs&, const Knobs&, const Workspace&, hipStream_t); \
    hipError_t ub_compact_h##h(int, int, const CompactArgs&, const Knobs&, const Workspace&, hipStream_t); \
    int64_t lane_rec_len_h##h(int dtype);                                                          \
    const char* lane_build_h##h();
TPC_DECL_H(4) TPC_DECL_H(5) TPC_DECL_H(10) TPC_DECL_H(20) TPC_DECL_H(30) TPC_DECL_H(40)
#undef TPC_DECL_H
// ... and for the general model (mpc_groupg_inst.hip)
#define TPC_DECL_H(h) hipError_t groupg_general_h##h(int, int, int, const GeneralArgs&, const Knobs&, const Workspace&, hipStream_t);
TPC_DECL_H(10) TPC_DECL_H(20) TPC_DECL_H(30) TPC_DECL_H(40)
#undef TPC_DECL_H
// LANE_FMA for the general model (mpc_ubg_inst.hip): N <= 20
#define TPC_DECL_H(h) hipError_t ub_general_h##h(int, int, const GeneralArgs&, const Knobs&, const Workspace&, hipStream_t);
TPC_DECL_H(4) TPC_DECL_H(5) TPC_DECL_H(10) TPC_DECL_H(20)
#undef TPC_DECL_H
// GROUP: G lanes per instance (mpc_group_inst.hip), compact form, the horizons a power of two divides into chunks
#define TPC_DECL_H(h) hipError_t group_compact_h##h(int, int, int, const CompactArgs&, const Knobs&, const Workspace&, hipStream_t);
TPC_DECL_H(10) TPC_DECL_H(20) TPC_DECL_H(30) TPC_DECL_H(40)
#undef TPC_DECL_H

thread_local char g_create_error[kTpcErrLen] = "";
}  // namespace tpc

using namespace tpc;

static const int kHorizons[] = {4, 5, 10, 20, 30, 40};

namespace {

// horizons with specialised kernels; every other 1 <= H <= kMaxHorizon takes the generic kernel
bool horizon_specialised(int H) {
    for (int h : kHorizons) if (h == H) return true;
    return false;
}
bool horizon_ok(int H) { return H >= 1 && H <= kMaxHorizon; }

Knobs knobs_of(const tpc_mpc_params* p) {
    Knobs k;
    k.eps = p->eps;
    k.max_iter = (uint32_t)p->max_iter;
    k.smo_iters = (uint32_t)p->smo_iters;
    return k;
}

// One-shot: the hint set for this handle applies to the next batch solve of the same size only.
const int32_t* take_hint(tpc_mpc_context* h, int64_t n) {
    const int32_t* p = (h->hint && h->hint_n == n) ? h->hint : nullptr;
    h->hint = nullptr;
    h->hint_n = 0;
    return p;
}

// the host path's translation unit is built with -mfma: only on a CPU that has the instruction
bool host_path_usable() {
#if defined(__x86_64__)
    static const bool ok = __builtin_cpu_supports("fma");
    return ok;
#else
    return true;   // (aarch64: fused multiply-add is baseline)
#endif
}

// Lanes per instance of the GROUP family for horizon H: what TPC_MPC_OPT_GROUP_LANES pins if it divides H, else the
// measured best; 0 where the family has no kernel (N = 4, 5 and the non-specialised horizons).
int group_lanes(const tpc_mpc_context* h, int H, int dtype = 0, int64_t n = 0, int form = 0) {
    if (H != 10 && H != 20 && H != 30 && H != 40) return 0;
    // (what mpc_group_inst.hip / mpc_groupg_inst.hip build: chunks of 3 .. 20 steps -- 3 .. 10 for the general model --
    // padded where G does not divide H)
    auto built = [&](int G) {
        if (H == 10) return G == 2 || G == 4;
        if (form == 1 && H >= 30) return G == 4 || G == 8;
        return G == 2 || G == 4 || G == 8;
    };
    if (form == 1) dtype = TPC_MPC_F64;   // (the general form's rows were measured in fp64)
    const int want = h->opt_group_lanes;
    if (want > 0 && built(want)) return want;
    // the measured best for this batch size (auto_table.h); where a row names a size that is not built, the next one down
    for (const AutoRow& r : kAutoTable) {
        if (r.form != form || r.dtype != dtype || r.horizon != H || n <= 0) continue;
        auto scaled = [&](int64_t at) { return at * h->cu_count / 256 < at ? at * h->cu_count / 256 : at; };
        int g = n < scaled(r.group8_below) ? 8 : (n < scaled(r.group4_below) ? 4 : 2);
        while (g > 2 && !built(g)) g /= 2;
        if (!built(g)) g = 4;
        return g;
    }
    return H == 40 ? 8 : (H == 10 ? 2 : 4);
}

// GROUP, kernels built for more than one wavefront per SIMD (fp32, compact form): does a batch of n take the full grid?
// From auto_table.h's pair_from: below it the longest instance sets the time and a lone wavefront iterates faster.
bool group_pair(const tpc_mpc_context* h, int H, int dtype, int64_t n, int form) {
    for (const AutoRow& r : kAutoTable) {
        if (r.form != form || r.dtype != dtype || r.horizon != H) continue;
        const int64_t at = r.pair_from * h->cu_count / 256 < r.pair_from ? r.pair_from * h->cu_count / 256 : r.pair_from;
        return n >= at;
    }
    return false;
}

// LANE needs enough instances to give every SIMD a full wavefront; below that WAVE's
// one-wavefront-per-instance launch finishes sooner.  Measured crossovers on MI355X
// (scripts/crossover.py, fp64, compact form, round 2 kernels; the chip has 65 536 LANE slots):
// fp32 at H = 4 and H = 10 (one instance per wavefront): between 24 576 and 32 768 instances; H = 20: WAVE still
// wins at 32 768, the largest batch its work queue takes (5.4 against 6.9 ms), and loses without the queue beyond.
// The WAVE kernel maps one decision variable to one lane, so it exists for I*H <= 64 only.
// Returns the kernel family to run, or -1 when WAVE was demanded for a shape it cannot take.
// `fma_ok`: the request is one the LANE_FMA family takes (compact form with usable bounds: fma_usable(); general form:
// fma_general_usable()); `compact`: the compact form (its WAVE / LANE_FMA crossovers were measured separately).
int pick_algo(const tpc_mpc_context* h, int algo, int I, int H, int64_t n, int dtype, bool fma_ok = false,
              bool compact = true, bool group_general = false) {
    if (!horizon_specialised(H)) return algo == TPC_MPC_ALGO_WAVE ? -1 : kAlgoGeneric;
    const bool wave_ok = I * H <= kWave || (I == 2 && H <= kWave);   // (two variables per lane past 64)
    const int lane = fma_ok ? TPC_MPC_ALGO_LANE_FMA : TPC_MPC_ALGO_LANE;   // the throughput family of AUTO
    if (algo == TPC_MPC_ALGO_WAVE) return wave_ok ? algo : -1;
    if (algo == TPC_MPC_ALGO_LANE) return algo;
    // LANE_FMA named at N = 30 / 40, compact form, fp64: GROUP (the same arithmetic, G lanes per instance) takes the request.
    // The one-lane kernels there parked their state in scratch around every refill pass -- 0.28 / 4.4 GB of HBM traffic per
    // 262 144-instance launch, 26x / 419x the algorithmic bytes -- and lost to GROUP at every batch size (43.9 against 54-56 ms
    // at N = 40); they are no longer built (mpc_ub_inst.hip).  fp32 keeps them (no spill), as does the exact-stop-test build
    // GROUP itself falls back on.
    if (algo == TPC_MPC_ALGO_LANE_FMA && compact && fma_ok && dtype == TPC_MPC_F64 && H >= 30 && group_lanes(h, H) > 0)
        return TPC_MPC_ALGO_GROUP;
    if (algo == TPC_MPC_ALGO_LANE_FMA) return lane;
    // GROUP is LANE_FMA's arithmetic with G lanes per instance: the same requests, the horizons a group divides
    if (algo == TPC_MPC_ALGO_GROUP) return (compact ? (fma_ok && group_lanes(h, H) > 0) : group_general) ? algo : lane;
    const int64_t lanes = (int64_t)h->cu_count * 4 * kWave;
    // (N = 40 with two inputs: the two-variables-per-lane WAVE kernel against a LANE pass that lasts as long
    // as its slowest instance, 50 ms whatever the batch: 42.0 against 50.0 ms at 16 384, 62.4 against 49.9 at 24 576)
    // (fp64 with at most 32 variables: two instances per wavefront over the work queue -- WAVE wins up to the
    // largest batch the queue takes: 0.65 against 1.00 ms at 32 768 x N = 10, 0.20 against 0.21 at N = 4;
    // without the queue, at 49 152, it loses 1.6 to 1.0)
    const bool paired = dtype == TPC_MPC_F64 && I * H <= kWave / 2;
    // (the queue's crossover was measured on a 256-CU part; a smaller one -- a CPX partition -- scales it down)
    const int64_t queue_cross = (kWaveQueueMaxInstances * h->cu_count / 256 < kWaveQueueMaxInstances
                                     ? kWaveQueueMaxInstances * h->cu_count / 256 : kWaveQueueMaxInstances) + 1;
    // (N = 40, two inputs: a LANE pass lasts as long as its slowest instance whatever the batch -- 10 000 iterations of
    // 5.1 us in the bit-exact family, 2.9 us in LANE_FMA: 51 / 29.5 ms -- and the two-variables-per-lane WAVE kernel
    // with dense Hessian rows (the general form) takes 2.6 us per instance: 42 ms at 16 384.  So WAVE up to 19 456
    // instances against LANE ...)
    // ... and the compact model's WAVE kernel at N = 40 builds its gradient from prefix sums (mpc_wave.h,
    // scan_gradient): 0.92 us per instance through the queue, 15 ms at 16 384 -- ahead of either LANE family up to
    // the largest batch the queue takes.
    const int64_t two_per_lane_cross = (fma_ok && compact) ? queue_cross : lanes * 19 / 64;
    int64_t crossover = I * H > kWave ? two_per_lane_cross
                        : ((H >= 20 || paired) ? queue_cross : lanes * 7 / 16);
    if (fma_ok && compact) {
        // compact form: WAVE, then GROUP with fewer and fewer lanes per instance, then LANE_FMA, at crossovers measured
        // per dtype and horizon (auto_table.h, generated by scripts/measure_crossover.py; profiles/r04_crossover.txt).
        // N = 4 and 5 have no GROUP kernels (profiles/r03_crossover.txt: WAVE up to 28 672 in fp64, 24 576 / 21 504 in fp32).
        const bool d = dtype == TPC_MPC_F64;
        auto scaled = [&](int64_t at) { return at * h->cu_count / 256 < at ? at * h->cu_count / 256 : at; };
        for (const AutoRow& r : kAutoTable) {
            if (r.form != 0 || r.dtype != dtype || r.horizon != H) continue;
            if (n < scaled(r.wave_below) && n <= kWaveQueueMaxInstances) return TPC_MPC_ALGO_WAVE;
            if (n < scaled(r.group2_below)) return TPC_MPC_ALGO_GROUP;
            return lane;
        }
        int64_t at = kWaveQueueMaxInstances + 1;
        if (d && (H == 4 || H == 5)) at = 28672;
        if (!d && H == 4) at = 24576;
        if (!d && H == 5) at = 21504;
        crossover = scaled(at);
    }
    if (!compact && group_general) {
        // general form with group kernels: the same three-way split, from the rows measured for it (two inputs, fp64)
        auto scaled = [&](int64_t at) { return at * h->cu_count / 256 < at ? at * h->cu_count / 256 : at; };
        for (const AutoRow& r : kAutoTable) {
            if (r.form != 1 || r.horizon != H) continue;
            if (wave_ok && n < scaled(r.wave_below) && n <= kWaveQueueMaxInstances) return TPC_MPC_ALGO_WAVE;
            if (n < scaled(r.group2_below)) return TPC_MPC_ALGO_GROUP;
            return lane;
        }
    }
    return (n >= crossover || !wave_ok) ? lane : TPC_MPC_ALGO_WAVE;
}

// The unit-box coordinates of the LANE_FMA family need a box: finite bounds with upper > lower (dlib
// also accepts upper == lower, a pinned input: that goes to LANE).
bool fma_usable(const tpc_mpc_params* p) {
    for (int j = 0; j < 2; ++j)
        if (!(std::isfinite(p->lower[j]) && std::isfinite(p->upper[j]) && p->upper[j] > p->lower[j])) return false;
    return true;
}

// General form, GROUP (mpc_groupg_inst.hip): N = 10, 20, 30, 40; fp32 only where the general-form LANE_FMA unit stands
// in front of it (cold starts at N <= 20), fp64 also with the controller state in or out and at N = 30, 40.
bool group_general_usable(int dtype, int H, const void* controls, const void* v) {
    if (H != 10 && H != 20 && H != 30 && H != 40) return false;
    const bool via_lane = controls || v || H > 20;
    return !via_lane || dtype == TPC_MPC_F64;
}

// General form: the LANE_FMA kernels exist for N <= 20 and start cold (the controller state in or out is LANE's).
bool fma_general_usable(int H, const void* controls, const void* v) {
    return (H == 4 || H == 5 || H == 10 || H == 20) && !controls && !v;
}

int64_t lane_rec_len(int H, int dtype) {
    switch (H) {
#define X(h) case h: return lane_rec_len_h##h(dtype);
        X(4) X(5) X(10) X(20) X(30) X(40)
#undef X
    }
    return 0;
}

// the generic kernel is one launch: bracket it with the profiling events like the WAVE launcher does
template <class Launch> hipError_t generic_launch(const Workspace& ws, hipStream_t s, Launch launch) {
    if (ws.ev) (void)hipEventRecord(ws.ev[0], s);
    const hipError_t e = launch();
    if (ws.ev) { (void)hipEventRecord(ws.ev[1], s); (void)hipEventRecord(ws.ev[2], s); }
    return e;
}

hipError_t dispatch_compact(int algo, int H, int dtype, const CompactArgs& a, const Knobs& k,
                            const Workspace& ws, hipStream_t s) {
    if (algo == kAlgoGeneric) return generic_launch(ws, s, [&] { return generic_compact(dtype, H, a, k, ws.state, s); });
    if (algo == TPC_MPC_ALGO_LANE_FMA || algo == TPC_MPC_ALGO_GROUP) {
        // (compared in the arithmetic type: the kernels see the bounds rounded to it)
        const bool eqb = dtype == TPC_MPC_F64 ? (a.lo[0] == a.lo[1] && a.hi[0] == a.hi[1])
                                              : ((float)a.lo[0] == (float)a.lo[1] && (float)a.hi[0] == (float)a.hi[1]);
        if (algo == TPC_MPC_ALGO_GROUP) {
            switch (H) {
#define X(h) case h: return group_compact_h##h(dtype, eqb ? 1 : 0, ws.group_lanes, a, k, ws, s);
                X(10) X(20) X(30) X(40)
#undef X
            }
            return hipErrorInvalidValue;
        }
        switch (H) {
#define X(h) case h: return ub_compact_h##h(dtype, eqb ? 1 : 0, a, k, ws, s);
            X(4) X(5) X(10) X(20) X(30) X(40)
#undef X
        }
        return hipErrorInvalidValue;
    }
    switch (H) {
#define X(h) case h: return algo == TPC_MPC_ALGO_LANE ? lane_compact_h##h(dtype, a, k, ws, s) \
                                                       : wave_compact_h##h(dtype, a, k, ws, s);
        X(4) X(5) X(10) X(20) X(30) X(40)
#undef X
    }
    return hipErrorInvalidValue;
}
hipError_t dispatch_general(int algo, int I, int H, int dtype, const GeneralArgs& a, const Knobs& k,
                            const Workspace& ws, hipStream_t s) {
    if (algo == kAlgoGeneric) return generic_launch(ws, s, [&] { return generic_general(dtype, I, H, a, k, ws.state, s); });
    if (algo == TPC_MPC_ALGO_GROUP) {
        switch (H) {
#define X(h) case h: return groupg_general_h##h(dtype, I, ws.group_lanes, a, k, ws, s);
            X(10) X(20) X(30) X(40)
#undef X
        }
        return hipErrorInvalidValue;
    }
    if (algo == TPC_MPC_ALGO_LANE_FMA) {
        switch (H) {
#define X(h) case h: return ub_general_h##h(dtype, I, a, k, ws, s);
            X(4) X(5) X(10) X(20)
#undef X
        }
        return hipErrorInvalidValue;
    }
    switch (H) {
#define X(h) case h: return algo == TPC_MPC_ALGO_LANE ? lane_general_h##h(dtype, I, a, k, ws, s) \
                                                       : wave_general_h##h(dtype, I, a, k, ws, s);
        X(4) X(5) X(10) X(20) X(30) X(40)
#undef X
    }
    return hipErrorInvalidValue;
}

// AUTO's parity guarantee (include/tpc_mpc.h): where AUTO picked a tolerance family, the instances that family left on
// the iteration cap are solved once more by the bit-exact LANE arithmetic and published with dlib's bits.  fp64, the
// specialised horizons, cold starts; a host opts out with TPC_MPC_PARAM_FAST_CAPPED.
bool wants_cap_resolve(const tpc_mpc_params* p, int algo_ran) {
    return p->algo == TPC_MPC_ALGO_AUTO && p->dtype == TPC_MPC_F64 && p->max_iter > 0 &&
           (p->options & TPC_MPC_PARAM_FAST_CAPPED) == 0 && horizon_specialised(p->horizon) &&
           (algo_ran == TPC_MPC_ALGO_WAVE || algo_ran == TPC_MPC_ALGO_LANE_FMA || algo_ran == TPC_MPC_ALGO_GROUP);
}
// where the first pass leaves its iteration counts when the caller did not ask for them
int cap_iters_buffer(tpc_mpc_context* h, int64_t n, int32_t** out) {
    int rc = ensure(h, &h->cap_iters, &h->cap_iters_bytes, pad256(n * 4));
    if (rc) return rc;
    *out = (int32_t*)h->cap_iters;
    return TPC_MPC_OK;
}
hipError_t resolve_compact(int H, const CompactArgs& a, const Knobs& k, const Workspace& ws, hipStream_t s, double presolved_from = 0.0,
                           const uint32_t* pre_len = nullptr, uint32_t pre_limit = 0u) {
    switch (H) {
#define X(h) case h: return lane_resolve_compact_h##h(a, k, ws, a.iters, a.flags, s, presolved_from, pre_len, pre_limit);
        X(4) X(5) X(10) X(20) X(30) X(40)
#undef X
    }
    return hipErrorInvalidValue;
}
hipError_t resolve_general(int I, int H, const GeneralArgs& a, const Knobs& k, const Workspace& ws, hipStream_t s,
                           double presolved_from = 0.0, const uint32_t* pre_len = nullptr, uint32_t pre_limit = 0u) {
    switch (H) {
#define X(h) case h: return lane_resolve_general_h##h(I, a, k, ws, a.iters, a.flags, s, presolved_from, pre_len, pre_limit);
        X(4) X(5) X(10) X(20) X(30) X(40)
#undef X
    }
    return hipErrorInvalidValue;
}

int prepare_workspace(tpc_mpc_context* h, int algo, int H, int dtype, int64_t n, Workspace* ws, int form = 0) {
    ws->state = nullptr;
    ws->ticket = h->ws_words;
    ws->stats = (unsigned long long*)(h->ws_words + 4);
    ws->capacity_bytes = 0;
    ws->ev = h->profiling ? h->ev : nullptr;
    ws->wave_group = h->opt_wave_group;
    ws->group_lanes = group_lanes(h, H, dtype, n, form);
    ws->max_waves = h->max_waves;
    if (h->pre_busy && h->pre_group_waves > 0 && (ws->max_waves == 0 || ws->max_waves > h->pre_group_waves))
        ws->max_waves = h->pre_group_waves;   // a presolve holds the other SIMDs (presolve_begin)
    ws->group_pair = group_pair(h, H, dtype, n, form);
    ws->lanex_below = h->opt_lanex_below;
    ws->cu_count = h->cu_count;
    h->ev_valid = h->profiling;
    h->last_algo = algo;
    ws->keys = ws->rank = ws->order = nullptr;
    ws->sort_temp = nullptr;
    ws->sort_temp_bytes = 0;
    if (algo == kAlgoGeneric) {
        int rc = ensure(h, &h->ws_state, &h->ws_bytes, pad256(generic_scratch_bytes(H, dtype, n)));
        if (rc) return rc;
        ws->state = h->ws_state;
        ws->capacity_bytes = h->ws_bytes;
    }
    if (algo == TPC_MPC_ALGO_WAVE) {
        // order[] of the longest-first queue (mpc_wave.h); batches that fit the chip at once do not use it
        // + its ticket words, one cache line each (kQueueTickets x 256 B)
        int rc = ensure(h, &h->ws_state, &h->ws_bytes, pad256(n * 4) + kQueueTicketBytes);
        if (rc) return rc;
        ws->order = (uint32_t*)h->ws_state;
        ws->ticket = (uint32_t*)((char*)h->ws_state + pad256(n * 4));
        ws->capacity_bytes = h->ws_bytes;
    }
    if (algo == TPC_MPC_ALGO_LANE || algo == TPC_MPC_ALGO_LANE_FMA || algo == TPC_MPC_ALGO_GROUP) {
        // records | keys | rank | order | counting-sort bins
        const int64_t rec_b = pad256(lane_rec_len(H, dtype) * (int64_t)esize(dtype) * n);
        const int64_t col_b = pad256(n * 4);
        const size_t tmp_b = sort_temp_bytes(n);
        int rc = ensure(h, &h->ws_state, &h->ws_bytes, rec_b + 3 * col_b + pad256((int64_t)tmp_b));
        if (rc) return rc;
        char* b = (char*)h->ws_state;
        ws->state = b;
        ws->keys = (uint32_t*)(b + rec_b);
        ws->rank = (uint32_t*)(b + rec_b + col_b);
        ws->order = (uint32_t*)(b + rec_b + 2 * col_b);
        ws->sort_temp = b + rec_b + 3 * col_b;
        ws->sort_temp_bytes = tmp_b;
        ws->capacity_bytes = h->ws_bytes;
    }
    return TPC_MPC_OK;
}

// Workspace of the re-solve of capped instances (the LANE layout in the handle's scratch).  resolve_reserve() grows the
// scratch BEFORE the first pass is launched -- growing frees, and nothing may be freed under a running kernel's feet --
// resolve_workspace() carves it afterwards; ticket, statistics and events of the first pass stay what they are
// (tpc_mpc_last_lane_stats / _kernel_times describe the family that solved the batch).
int resolve_reserve(tpc_mpc_context* h, int H, int dtype, int64_t n) { return reserve_lane_workspace(h, H, dtype, n); }
int resolve_workspace(tpc_mpc_context* h, int H, int dtype, int64_t n, Workspace* ws) {
    const bool ev_valid = h->ev_valid;
    const int last_algo = h->last_algo;
    const int rc = prepare_workspace(h, TPC_MPC_ALGO_LANE, H, dtype, n, ws);
    h->ev_valid = ev_valid;
    h->last_algo = last_algo;
    ws->ev = nullptr;
    ws->ticket = h->ws_words + 16;   // ticket | queue length | statistics: 32 contiguous bytes (mpc_lane_inst.hip, resolve)
    ws->stats = (unsigned long long*)(h->ws_words + 18);
    return rc;
}

// Copy `rows` rows of `width` bytes between arrays of different leading dimensions (pitches in bytes).
hipError_t copy_rows(void* dst, int64_t dpitch, const void* src, int64_t spitch, int64_t width, int64_t rows,
                     hipMemcpyKind kind, hipStream_t s) {
    if (rows <= 0 || width <= 0) return hipSuccess;
    if (dpitch == width && spitch == width) return hipMemcpyAsync(dst, src, (size_t)(width * rows), kind, s);
    return hipMemcpy2DAsync(dst, (size_t)dpitch, src, (size_t)spitch, (size_t)width, (size_t)rows, kind, s);
}

int check_general_io(tpc_mpc_context* h, const tpc_mpc_general_io* io, int mem) {
    if (!io) return fail(h, TPC_MPC_ERR_BAD_ARG, "null io");
    if (io->inputs != 1 && io->inputs != 2) return fail(h, TPC_MPC_ERR_BAD_ARG, "inputs must be 1 or 2");
    if (io->n < 0 || io->ld < io->n || io->n > 0x7fffffffll)
        return fail(h, TPC_MPC_ERR_BAD_ARG, "need 0 <= n <= ld, n < 2^31");
    if (mem != TPC_MPC_HOST && mem != TPC_MPC_DEVICE) return fail(h, TPC_MPC_ERR_BAD_ARG, "bad memory kind");
    return TPC_MPC_OK;
}

}  // namespace

namespace tpc {

// A bin of a mixed-horizon batch that the GROUP family takes (tpc_mpc_mixed.hip runs those side by side, each on its
// share of the SIMDs): AUTO or GROUP asked for, bounds the unit box can take, a horizon with group kernels.
bool group_applicable(const tpc_mpc_context* h, const tpc_mpc_params* p, int H) {
    return (p->algo == TPC_MPC_ALGO_AUTO || p->algo == TPC_MPC_ALGO_GROUP) && fma_usable(p) && group_lanes(h, H) > 0;
}

// `host_ok`: the entry point also serves a host-only handle (tpc_mpc_solve_one); every other one needs the device
int check_common(tpc_mpc_context* h, const tpc_mpc_params* p, bool host_ok) {
    if (!h) return fail(nullptr, TPC_MPC_ERR_BAD_ARG, "null handle");
    if (h->host_only && !host_ok)
        return fail(h, TPC_MPC_ERR_NO_DEVICE, "host-only handle (TPC_MPC_DEVICE_NONE): only tpc_mpc_solve_one is served");
    if (!p) return fail(h, TPC_MPC_ERR_BAD_ARG, "null params");
    if (!horizon_ok(p->horizon))
        return fail(h, TPC_MPC_ERR_BAD_HORIZON, "horizon %d outside 1 .. %d", p->horizon, kMaxHorizon);
    if (p->dtype != TPC_MPC_F64 && p->dtype != TPC_MPC_F32) return fail(h, TPC_MPC_ERR_BAD_ARG, "bad dtype");
    if (p->algo < TPC_MPC_ALGO_AUTO || p->algo > TPC_MPC_ALGO_GROUP) return fail(h, TPC_MPC_ERR_BAD_ARG, "bad algo");
    if (!(p->eps > 0)) return fail(h, TPC_MPC_ERR_BAD_EPS, "eps must be > 0 (mpc.h:202)");
    if (p->options & ~TPC_MPC_PARAM_FAST_CAPPED) return fail(h, TPC_MPC_ERR_BAD_ARG, "unknown bits in tpc_mpc_params.options");
    if (p->max_iter > 0x7fffffffull || p->smo_iters > 0x7fffffffull)
        return fail(h, TPC_MPC_ERR_BAD_ARG, "max_iter / smo_iters must fit in 31 bits");
    return TPC_MPC_OK;
}

// mpc_abstract.h:90-97: min(Q) >= 0, min(R) > 0, min(upper-lower) >= 0
int check_compact_model(tpc_mpc_context* h, const tpc_mpc_params* p) {
    if (!(p->weight_y >= 0) || !(p->weight_phi >= 0))
        return fail(h, TPC_MPC_ERR_BAD_WEIGHTS, "min(Q) >= 0 violated (mpc_abstract.h:90-97)");
    if (!(p->weight_steering_front > 0) || !(p->weight_steering_rear > 0))
        return fail(h, TPC_MPC_ERR_BAD_WEIGHTS, "min(R) > 0 violated (mpc_abstract.h:90-97)");
    for (int j = 0; j < 2; ++j)
        if (!(p->upper[j] >= p->lower[j]))
            return fail(h, TPC_MPC_ERR_BAD_BOUNDS, "upper >= lower violated (mpc_abstract.h:90-97)");
    if (!std::isfinite(p->step_size) || !std::isfinite(p->wheelbase) || p->wheelbase == 0)
        return fail(h, TPC_MPC_ERR_BAD_ARG, "step_size / wheelbase must be finite, wheelbase != 0");
    return TPC_MPC_OK;
}

// StreamOrder.  Solves of one handle share its scratch (ticket, flag word, statistics, records,
// queue, staging), so two of them must never overlap.  On one stream they cannot; a solve that
// arrives on a different stream than the previous one first makes its stream wait for the event
// recorded behind that previous solve.
int stream_order_begin(tpc_mpc_context* h, hipStream_t s) {
    if (h->have_last && h->last_stream != s) HIP_TRY(h, hipStreamWaitEvent(s, h->done_ev, 0));
    return TPC_MPC_OK;
}
int stream_order_end(tpc_mpc_context* h, hipStream_t s) {
    HIP_TRY(h, hipEventRecord(h->done_ev, s));
    h->last_stream = s;
    h->have_last = true;
    return TPC_MPC_OK;
}

int finish_flags(tpc_mpc_context* h, uint32_t* flags_out, hipStream_t s) {
    if (!flags_out) return TPC_MPC_OK;
    uint32_t f = 0;
    HIP_TRY(h, hipMemcpyAsync(&f, h->ws_words + 1, sizeof(f), hipMemcpyDeviceToHost, s));
    HIP_TRY(h, hipStreamSynchronize(s));
    *flags_out = f;
    return TPC_MPC_OK;
}

int reserve_lane_workspace(tpc_mpc_context* h, int H, int dtype, int64_t n) {
    const bool ev_valid = h->ev_valid;
    const int last_algo = h->last_algo;
    Workspace ws;
    const int rc = prepare_workspace(h, horizon_specialised(H) ? TPC_MPC_ALGO_LANE : kAlgoGeneric, H, dtype, n, &ws);
    h->ev_valid = ev_valid;
    h->last_algo = last_algo;
    return rc;
}

int context_new(int device, int cu_count, tpc_mpc_context** out) {
    tpc_mpc_context* h = new (std::nothrow) tpc_mpc_context;
    if (!h) return fail(nullptr, TPC_MPC_ERR_ALLOC, "out of host memory");
    h->device = device;
    h->cu_count = cu_count;
    h->device_cu_count = cu_count;
    hipError_t e = hipSetDevice(device);
    if (e == hipSuccess) e = hipMalloc((void**)&h->ws_words, 128);
    if (e == hipSuccess) e = hipMemset(h->ws_words, 0, 128);
    if (e == hipSuccess) e = hipHostMalloc(&h->pin_host, 512, hipHostMallocMapped | hipHostMallocCoherent);
    if (e == hipSuccess) e = hipHostGetDevicePointer(&h->pin_dev, h->pin_host, 0);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&h->done_ev, hipEventDisableTiming);
    if (e != hipSuccess) {
        (void)tpc_mpc_destroy(h);
        return fail(nullptr, TPC_MPC_ERR_HIP, "create: %s", hipGetErrorString(e));
    }
    std::memset(h->pin_host, 0, 512);
    *out = h;
    return TPC_MPC_OK;
}

// The general form on DEVICE arrays, launches only (the sharded entry, tpc_mpc_comm.cpp, brackets it with the stream
// order, the flag word and the exchange): io describes the block to solve.
static double presolve_lambda(const tpc_mpc_params* p);
// AUTO's presolve (see presolve_begin below): scratch of its own in the LANE layout, `side_bytes` behind it, the side
// stream and its events
static int presolve_scratch(tpc_mpc_context* h, int H, int dtype, int64_t n, int64_t side_bytes, Workspace* ws, char** side) {
    const int64_t rec_b = pad256(lane_rec_len(H, dtype) * (int64_t)esize(dtype) * n), col_b = pad256(n * 4);
    const size_t tmp_b = sort_temp_bytes(n);
    const int64_t lane_b = rec_b + 3 * col_b + pad256((int64_t)tmp_b);
    int rc = ensure(h, &h->pre, &h->pre_bytes, lane_b + side_bytes);
    if (rc) return rc;
    if (!h->pre_stream) HIP_TRY(h, hipStreamCreateWithFlags(&h->pre_stream, hipStreamNonBlocking));
    if (!h->pre_fork) HIP_TRY(h, hipEventCreateWithFlags(&h->pre_fork, hipEventDisableTiming));
    if (!h->pre_done) HIP_TRY(h, hipEventCreateWithFlags(&h->pre_done, hipEventDisableTiming));
    char* b = (char*)h->pre;
    std::memset(ws, 0, sizeof(*ws));
    ws->state = b;
    ws->keys = (uint32_t*)(b + rec_b);
    ws->rank = (uint32_t*)(b + rec_b + col_b);
    ws->order = (uint32_t*)(b + rec_b + 2 * col_b);
    ws->sort_temp = b + rec_b + 3 * col_b;
    ws->sort_temp_bytes = tmp_b;
    ws->capacity_bytes = lane_b;
    ws->ev = nullptr;
    ws->lanex_below = h->opt_lanex_below;
    ws->cu_count = h->cu_count;
    ws->ticket = h->ws_words + 24;   // ticket | queue length | statistics: 32 contiguous bytes, as the second pass's (resolve_workspace)
    ws->stats = (unsigned long long*)(h->ws_words + 26);
    *side = b + lane_b;
    return TPC_MPC_OK;
}
// the share of the chip a presolve takes (see presolve_begin): false = this batch is not tried
static bool presolve_share(tpc_mpc_context* h, int64_t n, Presolve* ps) {
    const int simds = (h->device_cu_count > 0 ? h->device_cu_count : 256) * 4;
    const int solo_waves = simds * 7 / 16;
    if (n > (int64_t)solo_waves * 8 * 5) return false;
    ps->limit = (uint32_t)solo_waves * 8u;
    ps->group_waves = simds - solo_waves;
    return true;
}

int general_launch(tpc_mpc_context* h, const tpc_mpc_params* p, const tpc_mpc_general_io* io, hipStream_t s) {
    const int I = io->inputs, H = p->horizon;
    const int64_t n = io->n;
    const int algo = pick_algo(h, p->algo, I, H, n, p->dtype, fma_general_usable(H, io->controls_inout, io->v_inout), false,
                                   group_general_usable(p->dtype, H, io->controls_inout, io->v_inout));
    if (algo < 0) return fail(h, TPC_MPC_ERR_BAD_HORIZON, "the WAVE kernel exists for the specialised horizons with inputs*horizon <= 64 only; use LANE or AUTO");
    GeneralArgs a;
    std::memset(&a, 0, sizeof(a));
    a.n = n; a.ld = io->ld; a.shift_controls = 1;
    a.A = io->A; a.B = io->B; a.C = io->C; a.Q = io->Q; a.R = io->R; a.lo = io->lower; a.hi = io->upper;
    a.x0 = io->x0; a.targets = io->targets; a.controls = io->controls_inout; a.v = io->v_inout;
    a.u0 = io->u0; a.iters = io->iters;
    a.flags = h->ws_words + 1;
    a.work_hint = take_hint(h, n);
    const bool fix = wants_cap_resolve(p, algo) && !a.controls && !a.v;
    int rc = TPC_MPC_OK;
    if (fix && !a.iters) rc = cap_iters_buffer(h, n, &a.iters);
    if (!rc && fix) rc = resolve_reserve(h, H, p->dtype, n);
    if (rc) return rc;
    // AUTO's presolve, as the compact form's (presolve_begin): side outputs = I rows of u0 and the iteration counts
    Presolve ps;
    const double lf = presolve_lambda(p);
    const int64_t ubytes = pad256((int64_t)I * a.ld * 8);   // (the kernels address row j at u0 + j * ld: the side rows keep the caller's ld)
    if (fix && algo == TPC_MPC_ALGO_GROUP && lf > 0.0 && !h->pre_busy && presolve_share(h, n, &ps)) {
        Workspace wp;
        char* side = nullptr;
        rc = presolve_scratch(h, H, p->dtype, n, ubytes + pad256(n * 4), &wp, &side);
        if (rc) return rc;
        GeneralArgs a2 = a;
        a2.u0 = side;
        a2.iters = (int32_t*)(side + ubytes);
        a2.work_hint = nullptr;
        ps.side_front = side;
        ps.side_iters = a2.iters;
        ps.queue = wp.order;
        ps.queue_len = wp.ticket + 1;
        ps.lambda_from = lf;
        HIP_TRY(h, hipEventRecord(h->pre_fork, s));
        HIP_TRY(h, hipStreamWaitEvent(h->pre_stream, h->pre_fork, 0));
        ps.on = true;
        h->pre_busy = true;
        h->pre_group_waves = ps.group_waves;
        hipError_t e = hipErrorInvalidValue;
        switch (H) {
#define X(hh) case hh: e = lane_presolve_general_h##hh(I, a2, knobs_of(p), wp, lf, ps.limit, h->pre_stream); break;
            X(4) X(5) X(10) X(20) X(30) X(40)
#undef X
        }
        if (e != hipSuccess) {
            h->pre_busy = false;
            (void)hipEventRecord(h->pre_done, h->pre_stream);
            (void)hipStreamWaitEvent(s, h->pre_done, 0);
            return hip_fail(h, e, "kernel launch (presolve)");
        }
    }
    auto join = [&]() {
        if (!ps.on) return;
        ps.on = false;
        h->pre_busy = false;
        (void)hipEventRecord(h->pre_done, h->pre_stream);
        (void)hipStreamWaitEvent(s, h->pre_done, 0);
    };
    Workspace ws;
    rc = prepare_workspace(h, algo, H, p->dtype, n, &ws, 1);
    if (rc) { join(); return rc; }
    hipError_t e = dispatch_general(algo, I, H, p->dtype, a, knobs_of(p), ws, s);
    join();
    if (e == hipSuccess && ps.lambda_from > 0.0)
        e = presolve_merge_rows(ps.queue, ps.queue_len, n, ps.side_front, ps.side_iters, a.u0, a.iters, I, a.ld, ps.limit, s);
    if (e == hipSuccess && fix) {
        Workspace ws2;
        rc = resolve_workspace(h, H, p->dtype, n, &ws2);
        if (rc) return rc;
        e = resolve_general(I, H, a, knobs_of(p), ws2, s, ps.lambda_from, ps.queue_len, ps.limit);
    }
    if (e != hipSuccess) return hip_fail(h, e, "kernel launch");
    return TPC_MPC_OK;
}
int check_general_device_io(tpc_mpc_context* h, const tpc_mpc_general_io* io) {
    int rc = check_general_io(h, io, TPC_MPC_DEVICE);
    if (rc) return rc;
    if (io->n > 0 && (!io->A || !io->B || !io->C || !io->Q || !io->R || !io->lower || !io->upper || !io->x0 || !io->targets || !io->u0))
        return fail(h, TPC_MPC_ERR_BAD_ARG, "null batch pointer");
    return TPC_MPC_OK;
}

static void compact_args(tpc_mpc_context* h, const tpc_mpc_params* p, int64_t n, const void* v, const void* dy, const void* dphi,
                         void* front, void* rear, int32_t* iters, CompactArgs* a) {
    std::memset(a, 0, sizeof(*a));
    a->n = n;
    a->v = v; a->dy = dy; a->dphi = dphi;
    a->front = front; a->rear = rear; a->iters = iters;
    a->flags = h->collect_flags ? h->ws_words + 1 : nullptr;
    a->step = p->step_size; a->wheelbase = p->wheelbase;
    a->q[0] = p->weight_y; a->q[1] = p->weight_phi;
    a->r[0] = p->weight_steering_front; a->r[1] = p->weight_steering_rear;
    a->lo[0] = p->lower[0]; a->lo[1] = p->lower[1]; a->hi[0] = p->upper[0]; a->hi[1] = p->upper[1];
}

// Which instances will end on the iteration cap can be said before any of them is solved: dlib's iteration count grows like
// 5.5 sqrt(lambda) (lambda = its trace bound of the Hessian, mpc.h:116-123; measured with the checker on the synthetic
// streams: median 5.45; 6.36 at most among the instances near the cap at N = 40), so lambda >= (max_iter / 7)^2 names them --
// at N = 40 with dlib's default cap: every one of the 1 614 capped instances of a 16 384-instance batch, 20 % of the batch,
// half of those really capped.  The margin matters: ONE capped instance the prediction misses costs the second pass its
// whole 10 000-iteration chain again (with max_iter / 6.25 one of 16 384 slipped through: 31 ms instead of 16).  Still a
// prediction, not a decision: an instance it takes comes back with dlib's bits whether it capped or not, one it misses is
// caught by the flag-driven second pass as before.  Horizons below 30 never get near the cap with dlib's defaults and
// (and N = 30: lambda <= 1.1e6 at 4 m/s) are left alone: a presolve that finds nothing still costs GROUP its share of the chip.
double presolve_lambda_impl(const tpc_mpc_params* p) {
    if (p->horizon < 40 || p->max_iter < 2000) return 0.0;
    const double t = (double)p->max_iter / 7.0;
    return t * t;
}

static double presolve_lambda(const tpc_mpc_params* p) { return presolve_lambda_impl(p); }

int presolve_begin(tpc_mpc_context* h, const tpc_mpc_params* p, int64_t n, const void* v, const void* dy, const void* dphi,
                   hipStream_t s, Presolve* ps) {
    *ps = Presolve();
    const int algo = pick_algo(h, p->algo, 2, p->horizon, n, p->dtype, fma_usable(p));
    if (algo != TPC_MPC_ALGO_GROUP || !wants_cap_resolve(p, algo) || !h->collect_flags) return TPC_MPC_OK;
    const double lf = presolve_lambda(p);
    if (!(lf > 0.0) || h->pre_busy) return TPC_MPC_OK;
    // SIMDs of its own for the bit-exact kernel (mpc_lanex.h, SOLO), the rest for GROUP's persistent grid: 7 of 16 and 9 of
    // 16 -- the predicted set of the N = 40 stream is 20 % of a batch, eight instances per wavefront, and GROUP's pass is
    // bounded by its longest instance rather than by its share of the chip at the batch sizes it is AUTO's choice for
    // (16 384 x N = 40: 4.7 ms on the whole chip, 8.8 on 11 / 16).  A batch whose predicted set cannot run in ONE round of
    // those wavefronts is left to the second pass -- decided on the device, where the set's size is known; a batch too
    // large for a fifth of it to fit is not tried (presolve_share).
    if (!presolve_share(h, n, ps)) return TPC_MPC_OK;
    const int H = p->horizon;
    const int64_t out_b = pad256(n * 8), col_b = pad256(n * 4);
    Workspace ws;
    char* side = nullptr;
    int rc = presolve_scratch(h, H, p->dtype, n, 2 * out_b + col_b, &ws, &side);
    if (rc) return rc;
    char* b = side;
    const int64_t lane_b = 0;
    ps->side_front = b + lane_b;
    ps->side_rear = b + lane_b + out_b;
    ps->side_iters = (int32_t*)(b + lane_b + 2 * out_b);
    ps->queue = ws.order;
    ps->queue_len = ws.ticket + 1;
    ps->lambda_from = lf;
    CompactArgs a;
    compact_args(h, p, n, v, dy, dphi, ps->side_front, ps->side_rear, ps->side_iters, &a);
    HIP_TRY(h, hipEventRecord(h->pre_fork, s));
    HIP_TRY(h, hipStreamWaitEvent(h->pre_stream, h->pre_fork, 0));
    ps->on = true;   // (from here on presolve_finish must join the side stream, whatever happens)
    h->pre_busy = true;
    h->pre_group_waves = ps->group_waves;
    hipError_t e = hipErrorInvalidValue;
    switch (H) {
#define X(hh) case hh: e = lane_presolve_compact_h##hh(a, knobs_of(p), ws, lf, ps->limit, h->pre_stream); break;
        X(4) X(5) X(10) X(20) X(30) X(40)
#undef X
    }
    if (e != hipSuccess) return hip_fail(h, e, "kernel launch (presolve)");
    return TPC_MPC_OK;
}

int presolve_finish(tpc_mpc_context* h, const tpc_mpc_params* p, int64_t n, const void* v, const void* dy, const void* dphi,
                    void* front, void* rear, int32_t* iters, hipStream_t s, Presolve* ps) {
    if (ps->on) {
        ps->on = false;
        h->pre_busy = false;
        HIP_TRY(h, hipEventRecord(h->pre_done, h->pre_stream));
        HIP_TRY(h, hipStreamWaitEvent(s, h->pre_done, 0));
    }
    const int algo = pick_algo(h, p->algo, 2, p->horizon, n, p->dtype, fma_usable(p));
    if (algo < 0 || !wants_cap_resolve(p, algo) || !h->collect_flags) return TPC_MPC_OK;
    int32_t* sel = iters ? iters : ps->select;
    if (ps->lambda_from > 0.0) {
        hipError_t e = presolve_merge(ps->queue, ps->queue_len, n, ps->side_front, ps->side_rear, ps->side_iters, front, rear, sel, ps->limit, s);
        if (e != hipSuccess) return hip_fail(h, e, "kernel launch (presolve merge)");
    }
    CompactArgs a;
    compact_args(h, p, n, v, dy, dphi, front, rear, sel, &a);
    Workspace ws2;
    int rc = resolve_workspace(h, p->horizon, p->dtype, n, &ws2);
    if (rc) return rc;
    hipError_t e = resolve_compact(p->horizon, a, knobs_of(p), ws2, s, ps->lambda_from, ps->queue_len, ps->limit);
    if (e != hipSuccess) return hip_fail(h, e, "kernel launch");
    return TPC_MPC_OK;
}

// the tolerance (or whatever pick_algo says) pass; with `ps`: AUTO's guarantee is the caller's business (presolve_finish),
// immediately unless ps->deferred
int compact_launch_ps(tpc_mpc_context* h, const tpc_mpc_params* p, int64_t n, const void* v, const void* dy,
                      const void* dphi, void* front, void* rear, int32_t* iters, hipStream_t s, Presolve* ps) {
    const int algo = pick_algo(h, p->algo, 2, p->horizon, n, p->dtype, fma_usable(p));
    if (algo < 0) return fail(h, TPC_MPC_ERR_BAD_HORIZON, "the WAVE kernel exists for the specialised horizons with inputs*horizon <= 64 only; use LANE or AUTO");
    CompactArgs a;
    compact_args(h, p, n, v, dy, dphi, front, rear, iters, &a);
    a.work_hint = take_hint(h, n);
    const bool fix = wants_cap_resolve(p, algo) && a.flags;
    int rc = TPC_MPC_OK;
    if (fix && !a.iters) rc = cap_iters_buffer(h, n, &a.iters);
    if (!rc && fix) rc = resolve_reserve(h, p->horizon, p->dtype, n);
    if (rc) return rc;
    ps->select = a.iters;
    Workspace ws;
    rc = prepare_workspace(h, algo, p->horizon, p->dtype, n, &ws);
    if (rc) return rc;
    hipError_t e = dispatch_compact(algo, p->horizon, p->dtype, a, knobs_of(p), ws, s);
    if (e != hipSuccess) return hip_fail(h, e, "kernel launch");
    if (ps->deferred) return TPC_MPC_OK;
    return presolve_finish(h, p, n, v, dy, dphi, front, rear, iters, s, ps);
}

int compact_launch(tpc_mpc_context* h, const tpc_mpc_params* p, int64_t n, const void* v, const void* dy,
                   const void* dphi, void* front, void* rear, int32_t* iters, hipStream_t s) {
    Presolve ps;
    int rc = presolve_begin(h, p, n, v, dy, dphi, s, &ps);
    if (!rc) rc = compact_launch_ps(h, p, n, v, dy, dphi, front, rear, iters, s, &ps);
    if (rc && ps.on) {   // the side stream got work: the caller's stream must not run ahead of it (scratch, outputs)
        h->pre_busy = false;
        (void)hipEventRecord(h->pre_done, h->pre_stream);
        (void)hipStreamWaitEvent(s, h->pre_done, 0);
    }
    return rc;
}

}  // namespace tpc

extern "C" {

int tpc_mpc_abi_version(void) { return TPC_MPC_ABI_VERSION; }

int tpc_mpc_supported_horizons(int* out, int cap) {
    const int n = (int)(sizeof(kHorizons) / sizeof(kHorizons[0]));
    for (int i = 0; i < n && i < cap && out; ++i) out[i] = kHorizons[i];
    return n;
}

const char* tpc_mpc_build_info(void) {
    // assembled once from what each LANE translation unit recorded about its own build (csrc/Makefile);
    // static storage: valid for the process lifetime, as the header promises
    static char info[512];
    static std::once_flag once;
    std::call_once(once, [] {
        std::snprintf(info, sizeof(info),
                      "libtpc_mpc abi %d gfx950; lane units: h4[%s] h5[%s] h10[%s] h20[%s] h30[%s] h40[%s]",
                      TPC_MPC_ABI_VERSION, lane_build_h4(), lane_build_h5(), lane_build_h10(), lane_build_h20(),
                      lane_build_h30(), lane_build_h40());
    });
    return info;
}

int tpc_mpc_default_params(tpc_mpc_params* p, int horizon) {
    if (!p) return TPC_MPC_ERR_BAD_ARG;
    std::memset(p, 0, sizeof(*p));
    p->horizon = horizon;
    p->dtype = TPC_MPC_F64;
    p->algo = TPC_MPC_ALGO_AUTO;
    p->eps = 0.01;            // mpc.h:104
    p->max_iter = 10000;      // mpc.h:103
    p->smo_iters = 50;        // mpc.h:319
    p->step_size = 0.1;       // src/trajectory_point_follower.cpp:96
    p->wheelbase = 0.21;      // include/trajectory_point_follower.h:47
    p->weight_y = 20;         // src/trajectory_point_follower.cpp:92-95
    p->weight_phi = 7;
    p->weight_steering_front = 0.0005;
    p->weight_steering_rear = 10;
    const double alpha_max = 22 * M_PI / 180;   // src/trajectory_point_follower.cpp:16-18
    p->lower[0] = p->lower[1] = -alpha_max;
    p->upper[0] = p->upper[1] = alpha_max;
    return horizon_ok(horizon) ? TPC_MPC_OK : TPC_MPC_ERR_BAD_HORIZON;
}

int tpc_mpc_create(int device, tpc_mpc_handle* out) {
    return guarded(nullptr, [&]() -> int {
        if (!out) return fail(nullptr, TPC_MPC_ERR_BAD_ARG, "null out pointer");
        *out = nullptr;
        if (device == TPC_MPC_DEVICE_NONE) {   // host-only: no HIP call, here or ever
            tpc_mpc_context* h = new (std::nothrow) tpc_mpc_context;
            if (!h) return fail(nullptr, TPC_MPC_ERR_ALLOC, "out of host memory");
            h->device = -1;
            h->host_only = true;
            *out = h;
            return TPC_MPC_OK;
        }
        int count = 0;
        hipError_t e = hipGetDeviceCount(&count);
        if (e != hipSuccess || count <= 0)
            return fail(nullptr, TPC_MPC_ERR_NO_DEVICE, "no HIP device: %s",
                        e != hipSuccess ? hipGetErrorString(e) : "count == 0");
        if (device < 0 || device >= count) return fail(nullptr, TPC_MPC_ERR_NO_DEVICE, "device index out of range");
        hipDeviceProp_t prop;
        e = hipGetDeviceProperties(&prop, device);
        if (e != hipSuccess) return fail(nullptr, TPC_MPC_ERR_NO_DEVICE, "%s", hipGetErrorString(e));
        if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
            return fail(nullptr, TPC_MPC_ERR_NO_DEVICE, "device is %s, this library carries gfx950 code only",
                        prop.gcnArchName);
        return context_new(device, prop.multiProcessorCount, out);
    });
}

int tpc_mpc_destroy(tpc_mpc_handle h) {
    if (!h) return TPC_MPC_OK;
    return guarded(nullptr, [&]() -> int {
        if (h->host_only) { delete h; return TPC_MPC_OK; }
        (void)hipSetDevice(h->device);
        one_shot_destroy(h);   // stops the resident kernel before its mailbox goes away
        comm_destroy(h);
        for (int i = 0; i < tpc_mpc_context::kMaxKids; ++i) {
            if (h->kid_stream[i]) { (void)hipStreamSynchronize(h->kid_stream[i]); (void)hipStreamDestroy(h->kid_stream[i]); }
            if (h->kid_done[i]) (void)hipEventDestroy(h->kid_done[i]);
            if (h->kids[i]) (void)tpc_mpc_destroy(h->kids[i]);
        }
        if (h->fork_ev) (void)hipEventDestroy(h->fork_ev);
        if (h->ws_state) (void)hipFree(h->ws_state);
        if (h->ws_words) (void)hipFree(h->ws_words);
        if (h->stage) (void)hipFree(h->stage);
        if (h->roll) (void)hipFree(h->roll);
        if (h->cap_iters) (void)hipFree(h->cap_iters);
        if (h->mix) (void)hipFree(h->mix);
        if (h->gather) (void)hipFree(h->gather);
        if (h->pre_stream) { (void)hipStreamSynchronize(h->pre_stream); (void)hipStreamDestroy(h->pre_stream); }
        if (h->pre_fork) (void)hipEventDestroy(h->pre_fork);
        if (h->pre_done) (void)hipEventDestroy(h->pre_done);
        if (h->pre) (void)hipFree(h->pre);
        if (h->hint_own) (void)hipFree(h->hint_own);
        if (h->pin_host) (void)hipHostFree(h->pin_host);
        if (h->done_ev) (void)hipEventDestroy(h->done_ev);
        for (auto& e : h->ev) if (e) (void)hipEventDestroy(e);
        delete h;
        return TPC_MPC_OK;
    });
}

const char* tpc_mpc_last_error(tpc_mpc_handle h) { return h ? h->err : g_create_error; }

int tpc_mpc_solve_batch_compact(tpc_mpc_handle h, const tpc_mpc_params* p, int64_t n, const void* v,
                                const void* delta_y, const void* delta_phi, void* steering_front,
                                void* steering_rear, int32_t* iters, uint32_t* flags_out, int mem,
                                void* stream) {
    return guarded(h, [&]() -> int {
        int rc = check_common(h, p);
        if (rc) return rc;
        rc = check_compact_model(h, p);
        if (rc) return rc;
        if (n < 0 || n > 0x7fffffffll) return fail(h, TPC_MPC_ERR_BAD_ARG, "need 0 <= n < 2^31");
        if (mem != TPC_MPC_HOST && mem != TPC_MPC_DEVICE) return fail(h, TPC_MPC_ERR_BAD_ARG, "bad memory kind");
        if (n == 0) { if (flags_out) *flags_out = 0; return TPC_MPC_OK; }
        if (!v || !delta_y || !delta_phi || !steering_front || !steering_rear)
            return fail(h, TPC_MPC_ERR_BAD_ARG, "null batch pointer");
        HIP_TRY(h, hipSetDevice(h->device));
        hipStream_t s = (hipStream_t)stream;
        const int64_t es = (int64_t)esize(p->dtype);
        StreamOrderScope order(h, s);
        rc = order.begin();
        if (rc) return rc;
        HIP_TRY(h, hipMemsetAsync(h->ws_words + 1, 0, sizeof(uint32_t), s));
        if (mem == TPC_MPC_DEVICE) {
            rc = compact_launch(h, p, n, v, delta_y, delta_phi, steering_front, steering_rear, iters, s);
            if (rc) return rc;
        } else {
            // staging layout: v | dy | dphi | front | rear | iters
            const int64_t col = pad256(n * es), icol = pad256(n * 4);
            rc = ensure(h, &h->stage, &h->stage_bytes, 5 * col + icol);
            if (rc) return rc;
            char* b = (char*)h->stage;
            HIP_TRY(h, hipMemcpyAsync(b, v, n * es, hipMemcpyHostToDevice, s));
            HIP_TRY(h, hipMemcpyAsync(b + col, delta_y, n * es, hipMemcpyHostToDevice, s));
            HIP_TRY(h, hipMemcpyAsync(b + 2 * col, delta_phi, n * es, hipMemcpyHostToDevice, s));
            int32_t* d_it = iters ? (int32_t*)(b + 5 * col) : nullptr;
            rc = compact_launch(h, p, n, b, b + col, b + 2 * col, b + 3 * col, b + 4 * col, d_it, s);
            if (rc) return rc;
            HIP_TRY(h, hipMemcpyAsync(steering_front, b + 3 * col, n * es, hipMemcpyDeviceToHost, s));
            HIP_TRY(h, hipMemcpyAsync(steering_rear, b + 4 * col, n * es, hipMemcpyDeviceToHost, s));
            if (iters) HIP_TRY(h, hipMemcpyAsync(iters, d_it, n * 4, hipMemcpyDeviceToHost, s));
            HIP_TRY(h, hipStreamSynchronize(s));
        }
        rc = order.end();
        if (rc) return rc;
        return finish_flags(h, flags_out, s);
    });
}

int tpc_mpc_solve_one(tpc_mpc_handle h, const tpc_mpc_params* p, double v, double delta_y,
                      double delta_phi, double* steering_front, double* steering_rear) {
    return guarded(h, [&]() -> int {
        if (!h) return fail(nullptr, TPC_MPC_ERR_BAD_ARG, "null handle");
        if (!steering_front || !steering_rear) return fail(h, TPC_MPC_ERR_BAD_ARG, "null output pointer");
        int rc = check_common(h, p, true);
        if (rc) return rc;
        rc = check_compact_model(h, p);
        if (rc) return rc;
        // the host path (csrc/tpc_mpc_host.cpp): a host-only handle always, a GPU handle up to the horizon its option names
        if (h->host_only || p->horizon <= h->opt_host_horizon) {
            const bool takes = p->dtype == TPC_MPC_F64 && (p->algo == TPC_MPC_ALGO_AUTO || p->algo == TPC_MPC_ALGO_LANE_FMA) &&
                               fma_usable(p) && horizon_specialised(p->horizon) && host_path_usable();
            if (takes) {
                int iters = 0;
                unsigned flags = 0;
                if (host_solve_one(p, v, delta_y, delta_phi, steering_front, steering_rear, &iters, &flags) == 0) {
                    h->one_flags = flags;
                    h->one_iters = iters;
                    h->one_valid = true;
                    // AUTO's guarantee for a solve that ended on the cap: dlib's bits need the bit-exact LANE kernels
                    if ((flags & TPC_MPC_FLAG_MAX_ITER) && !h->host_only && p->algo == TPC_MPC_ALGO_AUTO && p->max_iter > 0 &&
                        (p->options & TPC_MPC_PARAM_FAST_CAPPED) == 0) {
                        tpc_mpc_params q = *p;
                        q.algo = TPC_MPC_ALGO_LANE;
                        return one_shot_solve(h, &q, v, delta_y, delta_phi, steering_front, steering_rear);
                    }
                    return TPC_MPC_OK;
                }
            }
            if (h->host_only)
                return fail(h, TPC_MPC_ERR_NO_DEVICE, "host-only handle: the host path takes fp64, AUTO or LANE_FMA, horizons 4 5 10 20 30 40, "
                                                      "finite bounds with upper > lower, on a CPU with fused multiply-add");
        }
        // (no HIP call on the resident path: the parts of one_shot_solve that launch or wait select the device themselves)
        return one_shot_solve(h, p, v, delta_y, delta_phi, steering_front, steering_rear);
    });
}

int tpc_mpc_last_flags(tpc_mpc_handle h, uint32_t* flags, int32_t* iters) {
    return guarded(h, [&]() -> int {
        if (!h) return fail(nullptr, TPC_MPC_ERR_BAD_ARG, "null handle");
        if (!h->one_valid) return fail(h, TPC_MPC_ERR_BAD_ARG, "no tpc_mpc_solve_one on this handle yet");
        if (flags) *flags = h->one_flags;
        if (iters) *iters = h->one_iters;
        return TPC_MPC_OK;
    });
}

int tpc_mpc_solve_batch_general(tpc_mpc_handle h, const tpc_mpc_params* p,
                                const tpc_mpc_general_io* io, uint32_t* flags_out, int mem,
                                void* stream) {
    return guarded(h, [&]() -> int {
        int rc = check_common(h, p);
        if (rc) return rc;
        rc = check_general_io(h, io, mem);
        if (rc) return rc;
        if (io->n == 0) { if (flags_out) *flags_out = 0; return TPC_MPC_OK; }
        if (!io->A || !io->B || !io->C || !io->Q || !io->R || !io->lower || !io->upper || !io->x0 ||
            !io->targets || !io->u0)
            return fail(h, TPC_MPC_ERR_BAD_ARG, "null batch pointer");
        HIP_TRY(h, hipSetDevice(h->device));
        hipStream_t s = (hipStream_t)stream;
        const int64_t es = (int64_t)esize(p->dtype);
        const int I = io->inputs, H = p->horizon;
        const int64_t n = io->n;
        const int algo = pick_algo(h, p->algo, I, H, n, p->dtype, fma_general_usable(H, io->controls_inout, io->v_inout), false,
                                   group_general_usable(p->dtype, H, io->controls_inout, io->v_inout));
        if (algo < 0) return fail(h, TPC_MPC_ERR_BAD_HORIZON, "the WAVE kernel exists for the specialised horizons with inputs*horizon <= 64 only; use LANE or AUTO");
        StreamOrderScope order(h, s);
        rc = order.begin();
        if (rc) return rc;

        GeneralArgs a;
        std::memset(&a, 0, sizeof(a));
        a.n = n;
        a.shift_controls = 1;
        // HOST arrays: component c of the caller's array starts at base + c*ld and only its first n
        // elements belong to this call (a shard passes base + k0 and the full batch's ld), so every
        // component row is copied on its own -- n elements, never ld -- into a staging array whose
        // leading dimension is n rounded up to a wavefront.
        const int64_t lds = (n + 63) / 64 * 64;
        const int comps[12] = {4, 2 * I, 2, 2, I, I, I, 2, 2 * H, H * I, H * I, I};
        int64_t off[13] = {0};
        char* b = nullptr;
        if (mem == TPC_MPC_DEVICE) {
            a.ld = io->ld;
            a.A = io->A; a.B = io->B; a.C = io->C; a.Q = io->Q; a.R = io->R; a.lo = io->lower; a.hi = io->upper;
            a.x0 = io->x0; a.targets = io->targets; a.controls = io->controls_inout; a.v = io->v_inout;
            a.u0 = io->u0; a.iters = io->iters;
        } else {
            const void* src[12] = {io->A, io->B, io->C, io->Q, io->R, io->lower, io->upper, io->x0, io->targets,
                                   io->controls_inout, io->v_inout, nullptr};
            int64_t total = 0;
            for (int c = 0; c < 12; ++c) { off[c] = total; total += pad256((int64_t)comps[c] * lds * es); }
            off[12] = total;
            total += pad256(n * 4);
            rc = ensure(h, &h->stage, &h->stage_bytes, total);
            if (rc) return rc;
            b = (char*)h->stage;
            for (int c = 0; c < 11; ++c)
                if (src[c])
                    HIP_TRY(h, copy_rows(b + off[c], lds * es, src[c], io->ld * es, n * es, comps[c],
                                         hipMemcpyHostToDevice, s));
            a.ld = lds;
            a.A = b + off[0]; a.B = b + off[1]; a.C = b + off[2]; a.Q = b + off[3]; a.R = b + off[4];
            a.lo = b + off[5]; a.hi = b + off[6]; a.x0 = b + off[7]; a.targets = b + off[8];
            a.controls = io->controls_inout ? b + off[9] : nullptr;
            a.v = io->v_inout ? b + off[10] : nullptr;
            a.u0 = b + off[11];
            a.iters = io->iters ? (int32_t*)(b + off[12]) : nullptr;
        }
        a.flags = h->ws_words + 1;
        a.work_hint = take_hint(h, n);
        const bool fix = wants_cap_resolve(p, algo) && !a.controls && !a.v;
        if (fix && !a.iters) {
            rc = cap_iters_buffer(h, n, &a.iters);
            if (rc) return rc;
        }
        if (fix) {
            rc = resolve_reserve(h, H, p->dtype, n);
            if (rc) return rc;
        }

        Workspace ws;
        rc = prepare_workspace(h, algo, H, p->dtype, n, &ws, 1);
        if (rc) return rc;
        HIP_TRY(h, hipMemsetAsync(h->ws_words + 1, 0, sizeof(uint32_t), s));
        hipError_t e = dispatch_general(algo, I, H, p->dtype, a, knobs_of(p), ws, s);
        if (e == hipSuccess && fix) {
            Workspace ws2;
            rc = resolve_workspace(h, H, p->dtype, n, &ws2);
            if (rc) return rc;
            e = resolve_general(I, H, a, knobs_of(p), ws2, s);
        }
        if (e != hipSuccess) return hip_fail(h, e, "kernel launch");
        if (mem == TPC_MPC_HOST) {
            const int64_t hp = io->ld * es, dp = lds * es, w = n * es;
            HIP_TRY(h, copy_rows(io->u0, hp, a.u0, dp, w, I, hipMemcpyDeviceToHost, s));
            if (io->controls_inout)
                HIP_TRY(h, copy_rows(io->controls_inout, hp, a.controls, dp, w, H * I, hipMemcpyDeviceToHost, s));
            if (io->v_inout) HIP_TRY(h, copy_rows(io->v_inout, hp, a.v, dp, w, H * I, hipMemcpyDeviceToHost, s));
            if (io->iters) HIP_TRY(h, hipMemcpyAsync(io->iters, a.iters, n * 4, hipMemcpyDeviceToHost, s));
            HIP_TRY(h, hipStreamSynchronize(s));
        }
        rc = order.end();
        if (rc) return rc;
        return finish_flags(h, flags_out, s);
    });
}

int tpc_mpc_rollout(tpc_mpc_handle h, const tpc_mpc_params* p, const tpc_mpc_general_io* io,
                    int32_t steps, const void* new_last_targets, void* controls_out,
                    void* states_out, int32_t* iters_out, uint32_t* flags_out, int mem,
                    void* stream) {
    return guarded(h, [&]() -> int {
        int rc = check_common(h, p);
        if (rc) return rc;
        rc = check_general_io(h, io, mem);
        if (rc) return rc;
        if (steps < 0 || steps > (1 << 24)) return fail(h, TPC_MPC_ERR_BAD_ARG, "need 0 <= steps <= 2^24");
        if (io->n == 0 || steps == 0) { if (flags_out) *flags_out = 0; return TPC_MPC_OK; }
        if (!io->A || !io->B || !io->C || !io->Q || !io->R || !io->lower || !io->upper || !io->x0 ||
            !io->targets || !controls_out)
            return fail(h, TPC_MPC_ERR_BAD_ARG, "null batch pointer");
        HIP_TRY(h, hipSetDevice(h->device));
        hipStream_t s = (hipStream_t)stream;
        const int64_t es = (int64_t)esize(p->dtype);
        const int I = io->inputs, H = p->horizon;
        const int64_t n = io->n, ld = io->ld;
        const int algo = pick_algo(h, p->algo, I, H, n, p->dtype, false, false, group_general_usable(p->dtype, H, (const void*)1, (const void*)1));
        if (algo < 0) return fail(h, TPC_MPC_ERR_BAD_HORIZON, "the WAVE kernel exists for the specialised horizons with inputs*horizon <= 64 only; use LANE or AUTO");
        StreamOrderScope order(h, s);
        rc = order.begin();
        if (rc) return rc;
        const bool host = mem == TPC_MPC_HOST;
        const hipMemcpyKind in_kind = host ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice;
        const hipMemcpyKind out_kind = host ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;

        // The handle's working set, leading dimension ldw (n rounded up to a wavefront): a copy of the
        // model, then the state the loop carries -- x, targets, controls, v -- and the per-step u0 and
        // iteration counts.  Copies are row-wise with n elements per component, so a caller's ld > n
        // (a shard of a larger batch) is neither over-read nor over-written.
        //   0 A[4] 1 B[2I] 2 C[2] 3 Q[2] 4 R[I] 5 lo[I] 6 hi[I] 7 x[2] 8 targets[2H] 9 controls[HI] 10 v[HI] 11 u0[I]
        const int64_t ldw = (n + 63) / 64 * 64;
        const int comps[12] = {4, 2 * I, 2, 2, I, I, I, 2, 2 * H, H * I, H * I, I};
        const void* src[12] = {io->A, io->B, io->C, io->Q, io->R, io->lower, io->upper, io->x0, io->targets,
                               io->controls_inout, io->v_inout, nullptr};
        int64_t off[12], total = 0;
        for (int c = 0; c < 12; ++c) { off[c] = total; total += pad256((int64_t)comps[c] * ldw * es); }
        const int64_t o_iters = total; total += pad256(ldw * 4);
        rc = ensure(h, &h->roll, &h->roll_bytes, total);
        if (rc) return rc;
        char* w = (char*)h->roll;
        for (int c = 0; c < 11; ++c) {
            if (src[c]) HIP_TRY(h, copy_rows(w + off[c], ldw * es, src[c], ld * es, n * es, comps[c], in_kind, s));
            else HIP_TRY(h, hipMemsetAsync(w + off[c], 0, (size_t)((int64_t)comps[c] * ldw * es), s));
        }
        // HOST mode: new_last_targets in, and the three per-step outputs, go through the staging buffer
        const void* d_nlt = new_last_targets;
        int64_t ld_nlt = ld, ld_out = ld;
        char *d_ctrl = (char*)controls_out, *d_states = (char*)states_out;
        int32_t* d_iters = iters_out;
        if (host) {
            const int64_t s_nlt = 0, s_ctrl = s_nlt + pad256((int64_t)steps * 2 * ldw * es);
            const int64_t s_states = s_ctrl + pad256((int64_t)steps * I * ldw * es);
            const int64_t s_iters = s_states + pad256((int64_t)steps * 2 * ldw * es);
            rc = ensure(h, &h->stage, &h->stage_bytes, s_iters + pad256((int64_t)steps * ldw * 4));
            if (rc) return rc;
            char* b = (char*)h->stage;
            if (new_last_targets) {
                HIP_TRY(h, copy_rows(b + s_nlt, ldw * es, new_last_targets, ld * es, n * es, (int64_t)steps * 2,
                                     hipMemcpyHostToDevice, s));
                d_nlt = b + s_nlt;
            }
            ld_nlt = ld_out = ldw;
            d_ctrl = b + s_ctrl;
            d_states = states_out ? b + s_states : nullptr;
            d_iters = iters_out ? (int32_t*)(b + s_iters) : nullptr;
        }

        GeneralArgs a;
        std::memset(&a, 0, sizeof(a));
        a.n = n; a.ld = ldw; a.shift_controls = 1;
        a.A = w + off[0]; a.B = w + off[1]; a.C = w + off[2]; a.Q = w + off[3]; a.R = w + off[4];
        a.lo = w + off[5]; a.hi = w + off[6];
        a.x0 = w + off[7]; a.targets = w + off[8]; a.controls = w + off[9]; a.v = w + off[10]; a.u0 = w + off[11];
        a.iters = (int32_t*)(w + o_iters);
        a.flags = h->ws_words + 1;

        RolloutStepArgs r;
        std::memset(&r, 0, sizeof(r));
        r.n = n; r.ld = ldw; r.ld_out = ld_out; r.ld_nlt = ld_nlt; r.I = I; r.H = H; r.steps = steps;
        r.A = a.A; r.B = a.B; r.C = a.C;
        r.x = w + off[7]; r.targets = w + off[8]; r.controls = w + off[9]; r.new_last_targets = d_nlt;
        r.controls_out = d_ctrl; r.states_out = d_states;
        r.iters_step = a.iters; r.iters_out = d_iters;

        Workspace ws;
        rc = prepare_workspace(h, algo, H, p->dtype, n, &ws, 1);
        if (rc) return rc;
        HIP_TRY(h, hipMemsetAsync(h->ws_words + 1, 0, sizeof(uint32_t), s));
        const Knobs kn = knobs_of(p);
        for (int st = 0; st < steps; ++st) {
            hipError_t e = dispatch_general(algo, I, H, p->dtype, a, kn, ws, s);
            if (e != hipSuccess) return hip_fail(h, e, "kernel launch");
            r.step = st;
            e = launch_rollout_step(p->dtype, r, s);
            if (e != hipSuccess) return hip_fail(h, e, "rollout step launch");
        }
        // controller state back to the caller
        if (io->controls_inout)
            HIP_TRY(h, copy_rows(io->controls_inout, ld * es, w + off[9], ldw * es, n * es, H * I, out_kind, s));
        if (io->v_inout) HIP_TRY(h, copy_rows(io->v_inout, ld * es, w + off[10], ldw * es, n * es, H * I, out_kind, s));
        if (host) {
            HIP_TRY(h, copy_rows(controls_out, ld * es, d_ctrl, ldw * es, n * es, (int64_t)steps * I, hipMemcpyDeviceToHost, s));
            if (states_out)
                HIP_TRY(h, copy_rows(states_out, ld * es, d_states, ldw * es, n * es, (int64_t)steps * 2, hipMemcpyDeviceToHost, s));
            if (iters_out)
                HIP_TRY(h, copy_rows(iters_out, ld * 4, d_iters, ldw * 4, n * 4, steps, hipMemcpyDeviceToHost, s));
            HIP_TRY(h, hipStreamSynchronize(s));
        }
        rc = order.end();
        if (rc) return rc;
        return finish_flags(h, flags_out, s);
    });
}

namespace {

int check_trajectories(tpc_mpc_context* h, const tpc_mpc_params* p, const tpc_mpc_trajectories* t,
                       const float* lookup_x, const float* lookup_y, int32_t lookup_n) {
    if (p->dtype != TPC_MPC_F64) return fail(h, TPC_MPC_ERR_BAD_ARG, "follow_batch solves in fp64");
    if (!t) return fail(h, TPC_MPC_ERR_BAD_ARG, "null trajectories");
    if (t->n < 0 || t->ld < t->n || t->n > 0x7fffffffll || t->max_points < 0 || lookup_n < 0)
        return fail(h, TPC_MPC_ERR_BAD_ARG, "need 0 <= n <= ld, n < 2^31, max_points >= 0, lookup_n >= 0");
    if (t->n == 0) return TPC_MPC_OK;
    if (!t->pos_x || !t->pos_y || !t->dir_x || !t->dir_y || !t->velocity || !t->count || !t->car_velocity ||
        !t->look_ahead || (lookup_n > 0 && (!lookup_x || !lookup_y)))
        return fail(h, TPC_MPC_ERR_BAD_ARG, "null batch pointer");
    return TPC_MPC_OK;
}

FollowArgs follow_args(const tpc_mpc_trajectories* t, const float* lookup_x, const float* lookup_y, int32_t lookup_n) {
    FollowArgs fa;
    std::memset(&fa, 0, sizeof(fa));
    fa.n = t->n; fa.ld = t->ld; fa.max_points = t->max_points;
    fa.px = t->pos_x; fa.py = t->pos_y; fa.dx = t->dir_x; fa.dy = t->dir_y; fa.vel = t->velocity;
    fa.count = t->count; fa.car_velocity = t->car_velocity; fa.look_ahead = t->look_ahead;
    fa.lut_x = lookup_x; fa.lut_y = lookup_y; fa.lut_n = lookup_n;
    return fa;
}

}  // namespace

int tpc_mpc_follow_batch(tpc_mpc_handle h, const tpc_mpc_params* p, const tpc_mpc_trajectories* t,
                         const float* lookup_x, const float* lookup_y, int32_t lookup_n,
                         double* steering_front, double* steering_rear, float* target_speed,
                         float* target_distance, int32_t* iters, uint32_t* flags_out, void* stream) {
    return guarded(h, [&]() -> int {
        int rc = check_common(h, p);
        if (rc) return rc;
        rc = check_compact_model(h, p);
        if (rc) return rc;
        rc = check_trajectories(h, p, t, lookup_x, lookup_y, lookup_n);
        if (rc) return rc;
        if (t->n == 0) { if (flags_out) *flags_out = 0; return TPC_MPC_OK; }
        if (!steering_front || !steering_rear || !target_speed || !target_distance)
            return fail(h, TPC_MPC_ERR_BAD_ARG, "null batch pointer");
        HIP_TRY(h, hipSetDevice(h->device));
        hipStream_t s = (hipStream_t)stream;
        const int64_t n = t->n;
        StreamOrderScope order(h, s);
        rc = order.begin();
        if (rc) return rc;
        // v | y_soll | phi_soll (the compact solve's inputs), produced on device
        const int64_t col = pad256(n * 8);
        rc = ensure(h, &h->roll, &h->roll_bytes, 3 * col);
        if (rc) return rc;
        char* b = (char*)h->roll;
        FollowArgs fa = follow_args(t, lookup_x, lookup_y, lookup_n);
        fa.v_out = (double*)b; fa.ysoll_out = (double*)(b + col); fa.phisoll_out = (double*)(b + 2 * col);
        fa.target_speed = target_speed; fa.target_distance = target_distance;
        hipError_t e = launch_traj_point(fa, s);
        if (e != hipSuccess) return hip_fail(h, e, "traj_point launch");
        HIP_TRY(h, hipMemsetAsync(h->ws_words + 1, 0, sizeof(uint32_t), s));
        rc = compact_launch(h, p, n, fa.v_out, fa.ysoll_out, fa.phisoll_out, steering_front, steering_rear, iters, s);
        if (rc) return rc;
        e = launch_follow_post(n, target_speed, steering_front, steering_rear, s);
        if (e != hipSuccess) return hip_fail(h, e, "follow_post launch");
        rc = order.end();
        if (rc) return rc;
        return finish_flags(h, flags_out, s);
    });
}

int tpc_mpc_follow_batch_horizon(tpc_mpc_handle h, const tpc_mpc_params* p, const tpc_mpc_trajectories* t,
                                 const float* step_spacing, const float* lookup_x, const float* lookup_y,
                                 int32_t lookup_n, double* steering_front, double* steering_rear,
                                 float* target_speed, float* target_distance, double* targets_out,
                                 int32_t* iters, uint32_t* flags_out, void* stream) {
    return guarded(h, [&]() -> int {
        int rc = check_common(h, p);
        if (rc) return rc;
        rc = check_compact_model(h, p);
        if (rc) return rc;
        rc = check_trajectories(h, p, t, lookup_x, lookup_y, lookup_n);
        if (rc) return rc;
        if (t->n == 0) { if (flags_out) *flags_out = 0; return TPC_MPC_OK; }
        if (!steering_front || !steering_rear || !target_speed || !target_distance)
            return fail(h, TPC_MPC_ERR_BAD_ARG, "null batch pointer");
        HIP_TRY(h, hipSetDevice(h->device));
        hipStream_t s = (hipStream_t)stream;
        const int64_t n = t->n;
        const int H = p->horizon, I = 2;
        const int algo = pick_algo(h, p->algo, I, H, n, p->dtype, fma_general_usable(H, nullptr, nullptr), false, group_general_usable(p->dtype, H, nullptr, nullptr));
        if (algo < 0) return fail(h, TPC_MPC_ERR_BAD_HORIZON, "the WAVE kernel exists for the specialised horizons with inputs*horizon <= 64 only; use LANE or AUTO");
        StreamOrderScope order(h, s);
        rc = order.begin();
        if (rc) return rc;
        // general-form batch built on device: A[4] B[4] C[2] Q[2] R[2] lo[2] hi[2] x0[2] targets[2H] u0[2]
        const int64_t ldw = (n + 63) / 64 * 64;
        const int comps[10] = {4, 4, 2, 2, 2, 2, 2, 2, 2 * H, 2};
        int64_t off[10], total = 0;
        for (int c = 0; c < 10; ++c) { off[c] = total; total += pad256((int64_t)comps[c] * ldw * 8); }
        rc = ensure(h, &h->roll, &h->roll_bytes, total);
        if (rc) return rc;
        char* w = (char*)h->roll;
        FollowArgs fa = follow_args(t, lookup_x, lookup_y, lookup_n);
        fa.target_speed = target_speed; fa.target_distance = target_distance;
        FollowHorizonArgs fh;
        std::memset(&fh, 0, sizeof(fh));
        fh.H = H; fh.ldw = ldw; fh.step_spacing = step_spacing;
        fh.step = p->step_size; fh.wheelbase = p->wheelbase;
        fh.q[0] = p->weight_y; fh.q[1] = p->weight_phi;
        fh.r[0] = p->weight_steering_front; fh.r[1] = p->weight_steering_rear;
        for (int j = 0; j < 2; ++j) { fh.lo[j] = p->lower[j]; fh.hi[j] = p->upper[j]; }
        fh.A = (double*)(w + off[0]); fh.B = (double*)(w + off[1]); fh.C = (double*)(w + off[2]);
        fh.Q = (double*)(w + off[3]); fh.R = (double*)(w + off[4]); fh.lo_out = (double*)(w + off[5]);
        fh.hi_out = (double*)(w + off[6]); fh.x0 = (double*)(w + off[7]); fh.targets = (double*)(w + off[8]);
        fh.targets_copy = targets_out; fh.ld_copy = n;
        hipError_t e = launch_traj_horizon(fa, fh, s);
        if (e != hipSuccess) return hip_fail(h, e, "traj_horizon launch");

        GeneralArgs a;
        std::memset(&a, 0, sizeof(a));
        a.n = n; a.ld = ldw; a.shift_controls = 1;
        a.A = fh.A; a.B = fh.B; a.C = fh.C; a.Q = fh.Q; a.R = fh.R; a.lo = fh.lo_out; a.hi = fh.hi_out;
        a.x0 = fh.x0; a.targets = fh.targets; a.u0 = w + off[9]; a.iters = iters;
        a.flags = h->ws_words + 1;
        const bool fix = wants_cap_resolve(p, algo);
        if (fix && !a.iters) {
            rc = cap_iters_buffer(h, n, &a.iters);
            if (rc) return rc;
        }
        if (fix) {
            rc = resolve_reserve(h, H, TPC_MPC_F64, n);
            if (rc) return rc;
        }
        Workspace ws;
        rc = prepare_workspace(h, algo, H, TPC_MPC_F64, n, &ws, 1);
        if (rc) return rc;
        HIP_TRY(h, hipMemsetAsync(h->ws_words + 1, 0, sizeof(uint32_t), s));
        e = dispatch_general(algo, I, H, TPC_MPC_F64, a, knobs_of(p), ws, s);
        if (e == hipSuccess && fix) {
            Workspace ws2;
            rc = resolve_workspace(h, H, TPC_MPC_F64, n, &ws2);
            if (rc) return rc;
            e = resolve_general(I, H, a, knobs_of(p), ws2, s);
        }
        if (e != hipSuccess) return hip_fail(h, e, "kernel launch");
        // u0[2][ldw] -> (front, rear), then the crossing rule
        HIP_TRY(h, hipMemcpyAsync(steering_front, w + off[9], n * 8, hipMemcpyDeviceToDevice, s));
        HIP_TRY(h, hipMemcpyAsync(steering_rear, w + off[9] + ldw * 8, n * 8, hipMemcpyDeviceToDevice, s));
        e = launch_follow_post(n, target_speed, steering_front, steering_rear, s);
        if (e != hipSuccess) return hip_fail(h, e, "follow_post launch");
        rc = order.end();
        if (rc) return rc;
        return finish_flags(h, flags_out, s);
    });
}

int tpc_mpc_reserve(tpc_mpc_handle h, const tpc_mpc_params* p, int64_t n, int mem) {
    return guarded(h, [&]() -> int {
        int rc = check_common(h, p);
        if (rc) return rc;
        if (n < 0 || n > 0x7fffffffll) return fail(h, TPC_MPC_ERR_BAD_ARG, "need 0 <= n < 2^31");
        if (mem != TPC_MPC_HOST && mem != TPC_MPC_DEVICE) return fail(h, TPC_MPC_ERR_BAD_ARG, "bad memory kind");
        if (n == 0) return TPC_MPC_OK;
        HIP_TRY(h, hipSetDevice(h->device));
        // the larger of the kernel families' needs, so that every choice of AUTO is covered -- and the buffer the
        // re-solve of capped instances reads its iteration counts from when the caller passes none
        rc = reserve_lane_workspace(h, p->horizon, p->dtype, n);
        if (rc) return rc;
        if (p->algo == TPC_MPC_ALGO_AUTO && p->dtype == TPC_MPC_F64) {
            int32_t* unused = nullptr;
            rc = cap_iters_buffer(h, n, &unused);
            if (rc) return rc;
            // AUTO's presolve: its scratch, side stream and events (sized as presolve_begin / general_launch size them
            // for two inputs and ld = n)
            Presolve ps;
            if (presolve_lambda(p) > 0.0 && presolve_share(h, n, &ps)) {
                Workspace unused_ws;
                char* side = nullptr;
                rc = presolve_scratch(h, p->horizon, p->dtype, n, 2 * pad256(n * 8) + pad256(n * 4), &unused_ws, &side);
                if (rc) return rc;
            }
        }
        if (mem == TPC_MPC_HOST) {
            const int64_t col = pad256(n * (int64_t)esize(p->dtype)), icol = pad256(n * 4);
            rc = ensure(h, &h->stage, &h->stage_bytes, 5 * col + icol);
            if (rc) return rc;
        }
        return TPC_MPC_OK;
    });
}

int tpc_mpc_x_set_work_hint(tpc_mpc_handle h, const int32_t* hint, int64_t n, int mem) {
    return guarded(h, [&]() -> int {
        if (!h) return fail(nullptr, TPC_MPC_ERR_BAD_ARG, "null handle");
        if (h->host_only) return fail(h, TPC_MPC_ERR_NO_DEVICE, "host-only handle");
        h->hint = nullptr;
        h->hint_n = 0;
        if (!hint || n == 0) return TPC_MPC_OK;   // cleared
        if (n < 0 || n > 0x7fffffffll) return fail(h, TPC_MPC_ERR_BAD_ARG, "need 0 <= n < 2^31");
        if (mem != TPC_MPC_HOST && mem != TPC_MPC_DEVICE) return fail(h, TPC_MPC_ERR_BAD_ARG, "bad memory kind");
        // always the handle's own copy: no caller pointer outlives the call that received it
        HIP_TRY(h, hipSetDevice(h->device));
        int rc = ensure(h, &h->hint_own, &h->hint_own_bytes, n * 4);
        if (rc) return rc;
        // the previous solve may still be reading the old copy: wait for it (its stream-order event), then copy
        // synchronously -- a DEVICE hint must be complete when this is called
        if (h->have_last) HIP_TRY(h, hipEventSynchronize(h->done_ev));
        HIP_TRY(h, hipMemcpy(h->hint_own, hint, (size_t)n * 4,
                             mem == TPC_MPC_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice));
        h->hint = (const int32_t*)h->hint_own;
        h->hint_n = n;
        return TPC_MPC_OK;
    });
}

int tpc_mpc_x_set_group_share(tpc_mpc_handle h, int waves, int cu_count) {
    return guarded(h, [&]() -> int {
        if (!h) return fail(nullptr, TPC_MPC_ERR_BAD_ARG, "null handle");
        if (waves < 0 || cu_count < 0) return fail(h, TPC_MPC_ERR_BAD_ARG, "need waves >= 0, cu_count >= 0");
        h->max_waves = waves;
        h->cu_count = cu_count > 0 ? cu_count : h->device_cu_count;   // (0: back to the device's own)
        return TPC_MPC_OK;
    });
}

int tpc_mpc_x_set_lanex_below(tpc_mpc_handle h, int64_t below) {
    return guarded(h, [&]() -> int {
        if (!h) return fail(nullptr, TPC_MPC_ERR_BAD_ARG, "null handle");
        if (below < -1) return fail(h, TPC_MPC_ERR_BAD_ARG, "need below >= -1");
        h->opt_lanex_below = below;
        return TPC_MPC_OK;
    });
}

int tpc_mpc_set_option(tpc_mpc_handle h, int option, int64_t value) {
    return guarded(h, [&]() -> int {
        if (!h) return fail(nullptr, TPC_MPC_ERR_BAD_ARG, "null handle");
        switch (option) {
            case TPC_MPC_OPT_WAVE_GROUP:
                if (value != 0 && value != 1 && value != 2 && value != 4)
                    return fail(h, TPC_MPC_ERR_BAD_ARG, "TPC_MPC_OPT_WAVE_GROUP takes 0, 1, 2 or 4");
                h->opt_wave_group = (int)value;
                return TPC_MPC_OK;
            case TPC_MPC_OPT_GROUP_LANES:
                if (value != 0 && value != 2 && value != 4 && value != 8)
                    return fail(h, TPC_MPC_ERR_BAD_ARG, "TPC_MPC_OPT_GROUP_LANES takes 0, 2, 4 or 8");
                h->opt_group_lanes = (int)value;
                return TPC_MPC_OK;
            case TPC_MPC_OPT_HOST_SOLVE_ONE:
                if (value < 0 || value > kMaxHorizon) return fail(h, TPC_MPC_ERR_BAD_ARG, "TPC_MPC_OPT_HOST_SOLVE_ONE takes a horizon, 0 .. 64");
                h->opt_host_horizon = (int)value;
                return TPC_MPC_OK;
            case TPC_MPC_OPT_MAILBOX_HOST:
                if (h->host_only) return fail(h, TPC_MPC_ERR_NO_DEVICE, "host-only handle");
                HIP_TRY(h, hipSetDevice(h->device));
                one_shot_destroy(h);   // the next solve_one sets its mailbox up again, where this option says
                h->opt_mailbox_host = value != 0;
                return TPC_MPC_OK;
        }
        return fail(h, TPC_MPC_ERR_BAD_ARG, "unknown option %d", option);
    });
}

int tpc_mpc_set_profiling(tpc_mpc_handle h, int enable) {
    return guarded(h, [&]() -> int {
        if (!h) return fail(nullptr, TPC_MPC_ERR_BAD_ARG, "null handle");
        if (h->host_only) return fail(h, TPC_MPC_ERR_NO_DEVICE, "host-only handle");
        HIP_TRY(h, hipSetDevice(h->device));
        if (enable)
            for (auto& e : h->ev) if (!e) HIP_TRY(h, hipEventCreate(&e));
        h->profiling = enable != 0;
        h->ev_valid = false;
        return TPC_MPC_OK;
    });
}

int tpc_mpc_last_kernel_times(tpc_mpc_handle h, double* first_ms, double* second_ms, int* algo) {
    return guarded(h, [&]() -> int {
        if (!h) return fail(nullptr, TPC_MPC_ERR_BAD_ARG, "null handle");
        if (!h->ev_valid)
            return fail(h, TPC_MPC_ERR_BAD_ARG, h->last_algo == kAlgoMixed ? "the last solve was a mixed-horizon batch: its bins ran on child handles"
                                                                            : "no profiled solve on this handle yet");
        HIP_TRY(h, hipEventSynchronize(h->ev[2]));
        float a = 0, b = 0;
        HIP_TRY(h, hipEventElapsedTime(&a, h->ev[0], h->ev[1]));
        HIP_TRY(h, hipEventElapsedTime(&b, h->ev[1], h->ev[2]));
        if (first_ms) *first_ms = a;
        if (second_ms) *second_ms = b;
        if (algo) *algo = h->last_algo;
        return TPC_MPC_OK;
    });
}

int tpc_mpc_last_lane_stats(tpc_mpc_handle h, uint64_t* wave_iterations, uint64_t* refill_blocks) {
    return guarded(h, [&]() -> int {
        if (!h) return fail(nullptr, TPC_MPC_ERR_BAD_ARG, "null handle");
        if (h->host_only) return fail(h, TPC_MPC_ERR_NO_DEVICE, "host-only handle");
        if (h->last_algo == kAlgoMixed)
            return fail(h, TPC_MPC_ERR_BAD_ARG, "the last solve was a mixed-horizon batch: its bins ran on child handles");
        HIP_TRY(h, hipSetDevice(h->device));
        unsigned long long st[2] = {0, 0};
        // waits for the handle's last solve only (its stream-order event), not for the whole device
        if (h->have_last) HIP_TRY(h, hipEventSynchronize(h->done_ev));
        HIP_TRY(h, hipMemcpy(st, h->ws_words + 4, sizeof(st), hipMemcpyDeviceToHost));
        if (wave_iterations) *wave_iterations = st[0];
        if (refill_blocks) *refill_blocks = st[1];
        return TPC_MPC_OK;
    });
}

}  // extern "C"
