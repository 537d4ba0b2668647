"""Parity tests proper (`-m gpu`): the HIP path, called through the C ABI, against
  * the golden vectors generated from the REAL dlib::mpc (tests/golden/), and
  * the pinned CPU oracle on seeded inputs,
plus size-independent properties at BASELINE.json's full batch sizes.

Tolerances (stated once, used below):
  LANE kernels, fp64 : bit-exact (max |du| == 0) and identical iteration counts -- they execute
                       dlib's own operation sequence.
  WAVE kernel,  fp64 : |du| <= 1e-9 absolute (=> <= 1e-6 relative for every |u| >= 1e-3; bounds are
                       +-0.384) and identical iteration counts; observed ~1e-11.
  LANE kernels, fp32 : bit-exact against the SAME restatement compiled with every value typed
                       float (oracle/liboracle_mpc_f32.so) and identical iteration counts.  dlib has
                       no fp32 form, so this checks "dlib's operation sequence in fp32", nothing more.
  fp32 vs fp64       : a throughput / tolerance-sweep mode, not a parity mode (SURVEY.md section 0
                       fact 3): only loose sanity bounds are asserted, the histogram is reported.
"""
import os

import numpy as np
import pytest

from conftest import bits_equal, bits_equal32, load_golden

pytestmark = pytest.mark.gpu

WAVE_ATOL = 1e-9


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch


def _solver(H, algo, dtype="f64", **kw):
    from trajectory_controller_amd import MpcSolver
    return MpcSolver(horizon=H, device=0, dtype=dtype, algo=algo, **kw)


def _dev(torch, *arrs, dtype=None):
    return [torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0", dtype=dtype) for a in arrs]


def _soa(a):
    """AoS [n, ...] -> component-major [comps, n] (the ABI's layout)."""
    a = np.asarray(a)
    return np.ascontiguousarray(a.reshape(a.shape[0], -1).T)


# ---------------------------------------------------------------------------------------------
# compact form (the reference module's own call pattern) against real-dlib golden vectors

@pytest.mark.parametrize("H", [4, 5, 10, 20, 40])
@pytest.mark.parametrize("algo", ["lane", "wave"])
def test_compact_golden(torch_cuda, H, algo):
    torch = torch_cuda
    g = load_golden(f"compact_H{H}.npz")
    v, dy, dphi = _dev(torch, g["v"], g["dy"], g["dphi"])
    with _solver(H, algo) as s:
        f, r, it = s.solve_batch_compact(v, dy, dphi, want_iters=True)
        torch.cuda.synchronize()
        f, r, it = f.cpu().numpy(), r.cpu().numpy(), it.cpu().numpy()
        assert s.last_flags & 1 == 0
    if algo == "lane":
        assert bits_equal(f, g["front"]) and bits_equal(r, g["rear"])
    else:
        assert np.abs(f - g["front"]).max() <= WAVE_ATOL
        assert np.abs(r - g["rear"]).max() <= WAVE_ATOL
    known = g["iters_lb"] >= 0
    assert np.all(it[known] >= g["iters_lb"][known])


@pytest.mark.parametrize("H", [4, 5, 10, 20, 30, 40])
@pytest.mark.parametrize("algo", ["lane", "wave"])
def test_compact_vs_oracle_iters(torch_cuda, oracle, H, algo):
    """Fresh seeded inputs (not the fixture ones): outputs AND iteration counts against the oracle, at every
    specialised horizon (WAVE at N = 40 is the prefix-sum kernel)."""
    from trajectory_controller_amd.synth import compact_inputs
    torch = torch_cuda
    n = 3000 if H < 40 else 700   # not a multiple of 64: exercises the ragged last wavefront
    v, dy, dphi = compact_inputs(H, n, first=100000)
    of, orr, oit = oracle.solve_compact(H, v, dy, dphi, nthreads=8)
    tv, ty, tp = _dev(torch, v, dy, dphi)
    with _solver(H, algo) as s:
        f, r, it = s.solve_batch_compact(tv, ty, tp, want_iters=True)
        f, r, it = f.cpu().numpy(), r.cpu().numpy(), it.cpu().numpy()
    assert np.array_equal(it, oit)
    if algo == "lane":
        assert bits_equal(f, of) and bits_equal(r, orr)
    else:
        assert max(np.abs(f - of).max(), np.abs(r - orr).max()) <= WAVE_ATOL


@pytest.mark.parametrize("algo", ["lane", "wave"])
def test_compact_knobs(torch_cuda, algo):
    """eps = 0.05, max_iter = 300: the knobs commented out at src/trajectory_point_follower.cpp:374-375."""
    torch = torch_cuda
    g = load_golden("compact_knobs_H10.npz")
    v, dy, dphi = _dev(torch, g["v"], g["dy"], g["dphi"])
    with _solver(10, algo, eps=float(g["eps"]), max_iter=int(g["max_iter"])) as s:
        f, r, it = s.solve_batch_compact(v, dy, dphi, want_iters=True)
        f, r, it = f.cpu().numpy(), r.cpu().numpy(), it.cpu().numpy()
    assert it.max() <= 300
    if algo == "lane":
        assert bits_equal(f, g["front"]) and bits_equal(r, g["rear"])
    else:
        assert max(np.abs(f - g["front"]).max(), np.abs(r - g["rear"]).max()) <= WAVE_ATOL


@pytest.mark.parametrize("H", [4, 20])
@pytest.mark.parametrize("algo", ["lane", "wave"])
def test_compact_edge_cases(torch_cuda, H, algo):
    """NaN inputs, zero target, saturating targets, crawling and very fast speeds (host-memory path)."""
    from trajectory_controller_amd import FLAG_NONFINITE
    g = load_golden("compact_edge.npz")
    with _solver(H, algo) as s:
        f, r, it = s.solve_batch_compact(g["v"], g["dy"], g["dphi"], want_iters=True)
        flags = s.last_flags
    if algo == "lane":
        assert bits_equal(f, g[f"front_H{H}"]) and bits_equal(r, g[f"rear_H{H}"])
    else:
        assert np.nanmax(np.abs(f - g[f"front_H{H}"])) <= WAVE_ATOL
        assert np.nanmax(np.abs(r - g[f"rear_H{H}"])) <= WAVE_ATOL
    # NaN -> the untouched start point (0,0) at iteration 0, and the non-fatal flag is raised
    assert np.all(f[:4] == 0) and np.all(r[:4] == 0) and np.all(it[:4] == 0)
    assert flags & FLAG_NONFINITE


def test_empty_and_tiny_batches(torch_cuda):
    torch = torch_cuda
    with _solver(10, "auto") as s:
        f, r = s.solve_batch_compact(np.empty(0), np.empty(0), np.empty(0))
        assert f.shape == (0,) and r.shape == (0,)
        e = torch.empty(0, dtype=torch.float64, device="cuda:0")
        f, r = s.solve_batch_compact(e, e, e)
        assert f.numel() == 0
        f, r = s.solve_batch_compact(np.array([1.0]), np.array([0.1]), np.array([0.05]))
        assert abs(f[0] - 0.34964671107011402) < 1e-9 and abs(r[0] - 0.075655735449909028) < 1e-9


def test_solve_one_reference_samples(torch_cuda):
    """Sample outputs of the real reference recorded in SURVEY.md Appendix B (config 1: a single
    instance through solve_one, to 1e-12 as SURVEY.md 8d asks)."""
    samples = [(4, 1.0, 0.1, 0.05, 0.28258865451261717, 0.059891817493776013),
               (10, 1.0, 0.1, 0.05, 0.34964671107011402, 0.075655735449909028),
               (20, 2.0, -0.2, 0.1, -0.34544297733739515, -0.21765810887303699),
               (40, 0.5, 0.05, -0.3, -0.11433869614255573, 0.16123074991160702)]
    for H, v, dy, dphi, ef, er in samples:
        with _solver(H, "auto") as s:
            f, r = s.mpc_controller_tobi(v, dy, dphi)
        assert abs(f - ef) <= 1e-12 and abs(r - er) <= 1e-12, (H, f, r)


def test_bad_arguments(torch_cuda):
    from trajectory_controller_amd import MpcSolver, TpcMpcError
    with pytest.raises(TpcMpcError) as e:
        MpcSolver(horizon=65)                  # 1 .. 64 are accepted (generic kernel beyond the specialised ones)
    assert e.value.status == 4
    with MpcSolver(horizon=7, algo="wave") as s:
        with pytest.raises(TpcMpcError) as e:   # no WAVE kernel for a non-specialised horizon
            s.solve_batch_compact(np.array([1.0]), np.array([0.1]), np.array([0.05]))
        assert e.value.status == 4
    with MpcSolver(horizon=10) as s:
        one = (np.array([1.0]), np.array([0.1]), np.array([0.05]))
        for over, status in ((dict(weight_steering_rear=0.0), 2), (dict(weight_y=-1.0), 2),
                             (dict(lower=(0.5, 0.5), upper=(0.1, 0.1)), 3), (dict(eps=0.0), 5)):
            with pytest.raises(TpcMpcError) as e:
                s.solve_batch_compact(*one, **over)
            assert e.value.status == status, over
    # the scratch and scheduling entry points validate like the solves do
    import ctypes as C
    from trajectory_controller_amd import capi
    with MpcSolver(horizon=10) as s:
        lib, h, p = s._lib, s._h, s._params()
        assert lib.tpc_mpc_reserve(h, C.byref(p), -1, capi.DEVICE) != capi.OK
        assert lib.tpc_mpc_reserve(h, C.byref(p), 16, 7) != capi.OK
        assert lib.tpc_mpc_reserve(h, None, 16, capi.DEVICE) != capi.OK
        assert lib.tpc_mpc_reserve(h, C.byref(p), 0, capi.DEVICE) == capi.OK
        assert lib.tpc_mpc_reserve(h, C.byref(p), 100000, capi.HOST) == capi.OK
        hint = (C.c_int32 * 4)(1, 2, 3, 4)
        assert lib.tpc_mpc_x_set_work_hint(h, hint, -1, capi.HOST) != capi.OK
        assert lib.tpc_mpc_x_set_work_hint(h, hint, 4, 9) != capi.OK
        assert lib.tpc_mpc_x_set_work_hint(h, hint, 4, capi.HOST) == capi.OK
        assert lib.tpc_mpc_x_set_work_hint(h, None, 0, capi.HOST) == capi.OK      # clears
        assert lib.tpc_mpc_x_set_work_hint(None, hint, 4, capi.HOST) != capi.OK


# ---------------------------------------------------------------------------------------------
# general dlib::mpc<2,I,H> surface

@pytest.mark.parametrize("I", [1, 2])
@pytest.mark.parametrize("H", [4, 5, 10, 20, 30, 40])
@pytest.mark.parametrize("algo", ["lane", "wave"])
def test_general_golden(torch_cuda, I, H, algo):
    """Real-dlib fixtures for every horizon the library ships (cold start: the fused LANE kernels,
    and every WAVE shape: one variable per lane up to I*H = 64, two per lane at I = 2, H = 40)."""
    torch = torch_cuda
    g = load_golden(f"general_I{I}_H{H}.npz")
    names = ["A", "B", "C", "Q", "R", "lo", "hi", "x0", "targets"]
    dev = _dev(torch, *[_soa(g[k]) for k in names])
    with _solver(H, algo) as s:
        u0 = s.solve_batch_general(*dev, inputs=I)
        u0 = u0.cpu().numpy()
    if algo == "lane":
        assert bits_equal(u0.T, g["u0"])
    else:
        assert np.abs(u0.T - g["u0"]).max() <= WAVE_ATOL


@pytest.mark.parametrize("I", [1, 2])
def test_general_h40_vs_oracle(torch_cuda, oracle, I):
    """The largest LANE kernels (general form, H = 40: 512 registers plus scratch).  LLVM's
    iterative-ILP scheduler miscompiled the I = 2 ones into a wild address, so this horizon is built
    with the default scheduler (csrc/Makefile); this test is what watches that decision."""
    from trajectory_controller_amd.synth import general_inputs
    H, n = 40, 600
    g = general_inputs(H, n, I=I, first=77)
    u0, _, it = oracle.solve_general(I, H, g["A"], g["B"], g["C"], g["Q"], g["R"], g["lo"], g["hi"],
                                     g["x0"], g["targets"], nthreads=8)
    names = ["A", "B", "C", "Q", "R", "lo", "hi", "x0", "targets"]
    host = [_soa(g[k]) for k in names]
    with _solver(H, "lane") as s:
        gu0, git = s.solve_batch_general(*host, inputs=I, want_iters=True)
    assert np.array_equal(git, it)
    assert bits_equal(gu0.T, u0)


@pytest.mark.parametrize("I", [1, 2])
def test_general_h30_vs_oracle(torch_cuda, oracle, I):
    """H = 30 is the one horizon whose fused LANE kernel keeps its forward-pass array and momentum
    vector in AGPRs (FusedBig, mpc_lane.h); the general model also routes the linear term's
    intermediates through them.  Cold start, so the fused kernel runs."""
    from trajectory_controller_amd.synth import general_inputs
    H, n = 30, 1500
    g = general_inputs(H, n, I=I, first=31)
    u0, _, it = oracle.solve_general(I, H, g["A"], g["B"], g["C"], g["Q"], g["R"], g["lo"], g["hi"],
                                     g["x0"], g["targets"], nthreads=8)
    names = ["A", "B", "C", "Q", "R", "lo", "hi", "x0", "targets"]
    host = [_soa(g[k]) for k in names]
    with _solver(H, "lane") as s:
        gu0, git = s.solve_batch_general(*host, inputs=I, want_iters=True)
    assert np.array_equal(git, it)
    assert bits_equal(gu0.T, u0)


@pytest.mark.parametrize("algo", ["lane", "wave"])
def test_general_warm_start_vs_oracle(torch_cuda, oracle, algo):
    """controls_inout: the warm-start shift (mpc.h:231-232) and the full solved sequence."""
    from trajectory_controller_amd.synth import general_inputs
    I, H, n = 2, 10, 777
    g = general_inputs(H, n, I=I, first=4242)
    rng = np.random.default_rng(5)
    cin = rng.uniform(-0.3, 0.3, size=(n, H, I))
    u0, cout, it = oracle.solve_general(I, H, g["A"], g["B"], g["C"], g["Q"], g["R"], g["lo"], g["hi"],
                                        g["x0"], g["targets"], controls_in=cin, nthreads=8)
    names = ["A", "B", "C", "Q", "R", "lo", "hi", "x0", "targets"]
    host = [_soa(g[k]) for k in names]
    controls = _soa(cin)
    with _solver(H, algo) as s:
        gu0, git = s.solve_batch_general(*host, controls=controls, inputs=I, want_iters=True)
    assert np.array_equal(git, it)
    if algo == "lane":
        assert bits_equal(gu0.T, u0) and bits_equal(controls.T.reshape(n, H, I), cout)
    else:
        assert np.abs(gu0.T - u0).max() <= WAVE_ATOL
        assert np.abs(controls.T.reshape(n, H, I) - cout).max() <= WAVE_ATOL


# ---------------------------------------------------------------------------------------------
# closed loop: warm start + target shift (mpc.h:229-239)

@pytest.mark.parametrize("algo", ["lane", "wave"])
def test_rollout_kat(torch_cuda, algo):
    """The reference's own known-answer scenario (dlib_files/dlib/test/mpc.cpp:266-317):
    mpc<2,1,30>, eps 1e-8, 30 warm-started closed-loop steps, against what real dlib produced."""
    g = load_golden("rollout_kat.npz")
    n = 3   # three identical controllers: also checks instances do not interfere
    rep = lambda a: np.ascontiguousarray(np.repeat(np.asarray(a, dtype=np.float64).reshape(-1, 1), n, axis=1))
    with _solver(30, algo, eps=1e-8, max_iter=10000) as s:
        c, st, it = s.rollout(30, rep(g["A"]), rep(g["B"]), rep(g["C"]), rep(g["Q"]), rep(g["R"]),
                              rep(g["lo"]), rep(g["hi"]), rep(g["x0"]), rep(g["targets0"].reshape(-1)),
                              inputs=1, want_iters=True)
    for k in range(n):
        if algo == "lane":
            assert bits_equal(c[:, k], g["controls"][:, 0]) and bits_equal(st[:, k], g["states"].reshape(-1))
        else:
            # dlib's own acceptance threshold for this scenario is 1e-7 (test/mpc.cpp:312)
            assert np.abs(c[:, k] - g["controls"][:, 0]).max() <= 1e-7


@pytest.mark.parametrize("algo", ["lane", "wave"])
def test_rollout_i2(torch_cuda, algo):
    g = load_golden("rollout_I2_H10.npz")
    steps = int(g["steps"])
    col = lambda a: np.ascontiguousarray(np.asarray(a, dtype=np.float64).reshape(-1, 1))
    with _solver(10, algo) as s:
        c, st, _ = s.rollout(steps, col(g["A"]), col(g["B"]), col(g["C"]), col(g["Q"]), col(g["R"]),
                             col(g["lo"]), col(g["hi"]), col(g["x0"]), col(g["targets0"].reshape(-1)),
                             new_last_targets=col(g["new_last_targets"].reshape(-1)), inputs=2)
    if algo == "lane":
        assert bits_equal(c[:, 0], g["controls"].reshape(-1)) and bits_equal(st[:, 0], g["states"].reshape(-1))
    else:
        assert np.abs(c[:, 0] - g["controls"].reshape(-1)).max() <= 1e-7


# ---------------------------------------------------------------------------------------------
# BASELINE.json sizes: size-independent properties

@pytest.mark.parametrize("algo", ["lane", "auto"])
def test_full_size_properties(torch_cuda, algo):
    """Batch 262 144, N=20, fp64 (BASELINE config 3 in its parity-grade dtype), through the bit-exact LANE
    family and through AUTO (= LANE_FMA at this size): the first 1024 instances are the golden ones (bit for bit /
    within 1e-9); solving a permutation of the batch permutes the outputs bit-for-bit (no dependence on wave/lane
    placement or refill order); outputs obey the bounds; a re-run is bit-identical."""
    from trajectory_controller_amd.synth import compact_inputs
    torch = torch_cuda
    H, n = 20, 262144
    g = load_golden("compact_H20.npz")
    v, dy, dphi = compact_inputs(H, n)
    tv, ty, tp = _dev(torch, v, dy, dphi)
    with _solver(H, algo) as s:
        f, r, it = s.solve_batch_compact(tv, ty, tp, want_iters=True)
        f2, r2 = s.solve_batch_compact(tv, ty, tp)
        perm = torch.randperm(n, device="cuda:0", generator=torch.Generator(device="cuda:0").manual_seed(1))
        fp, rp = s.solve_batch_compact(tv[perm].contiguous(), ty[perm].contiguous(), tp[perm].contiguous())
        torch.cuda.synchronize()
    if algo == "lane":
        assert bits_equal(f[:1024].cpu().numpy(), g["front"]) and bits_equal(r[:1024].cpu().numpy(), g["rear"])
    else:
        assert np.abs(f[:1024].cpu().numpy() - g["front"]).max() <= 1e-9 and np.abs(r[:1024].cpu().numpy() - g["rear"]).max() <= 1e-9
    assert torch.equal(f, f2) and torch.equal(r, r2)
    assert torch.equal(f[perm], fp) and torch.equal(r[perm], rp)
    amax = 22 * np.pi / 180
    assert float(f.abs().max()) <= amax and float(r.abs().max()) <= amax
    it = it.cpu().numpy()
    assert it.min() >= 0 and it.max() <= 10000
    # iteration statistics of the reference on this distribution: mean ~1031
    assert 950 < it.mean() < 1100


@pytest.mark.parametrize("dtype", ["f32", "f64"])
def test_config4_total_batch_on_one_gpu(torch_cuda, dtype):
    """BASELINE config 4's WHOLE batch -- 2 097 152 trajectories, N=20, fp32 as written and fp64 -- on one GPU, and its
    eight rank blocks the way tpc_mpc_solve_batch_compact_sharded solves them (rank r: instances [r n, (r+1) n) into its
    slot of full-size outputs; the RCCL exchange, which this box cannot run at world 8, moves finished slots and
    changes no value).  Size-independent properties: within one kernel family every block's slot equals the whole
    solve's slice bit for bit (a result does not depend on which batch, queue position or rank an instance is solved
    in) -- fp64 under AUTO, which takes LANE_FMA at both sizes, fp32 with LANE_FMA named (AUTO takes GROUP for a
    262 144-instance fp32 block and LANE_FMA for the whole: the same arithmetic in another association, held to the
    fp32 tolerance statement below); fp64: the first 1024 instances are the real-dlib fixture within 1e-9; outputs
    obey the bounds; iteration counts are dlib's statistics."""
    from trajectory_controller_amd.synth import compact_inputs
    torch = torch_cuda
    H, world, n = 20, 8, 262144
    total = world * n
    tdt = torch.float32 if dtype == "f32" else torch.float64
    tv, ty, tp = _dev(torch, *compact_inputs(H, total), dtype=tdt)

    def whole_and_blocks(algo):
        with _solver(H, algo, dtype=dtype) as s:
            f, r, it = s.solve_batch_compact(tv, ty, tp, want_iters=True)
            assert s.last_flags & 1 == 0
            f_all, r_all = torch.full_like(f, float("nan")), torch.full_like(r, float("nan"))
            for rank in range(world):
                lo = rank * n
                # (the per-rank inputs of bench.py --gpus 8: compact_inputs(H, n, first=rank * n) is this slice)
                s.solve_batch_compact(tv[lo:lo + n], ty[lo:lo + n], tp[lo:lo + n], out=(f_all[lo:lo + n], r_all[lo:lo + n]))
            torch.cuda.synchronize()
        return f, r, it, f_all, r_all

    f, r, it, f_all, r_all = whole_and_blocks("auto" if dtype == "f64" else "lane_fma")
    assert torch.equal(f, f_all) and torch.equal(r, r_all)
    amax = 22 * np.pi / 180
    assert float(f.abs().max()) <= amax * (1 + 1e-6) and float(r.abs().max()) <= amax * (1 + 1e-6)
    it = it.cpu().numpy()
    assert it.min() >= 0 and it.max() <= 10000 and 950 < it.mean() < 1100
    if dtype == "f64":
        g = load_golden("compact_H20.npz")
        assert np.abs(f[:1024].cpu().numpy() - g["front"]).max() <= 1e-9
        assert np.abs(r[:1024].cpu().numpy() - g["rear"]).max() <= 1e-9
    else:
        # AUTO as a rank would run it: fp32 has no reference to be held to, so this is the tolerance statement of
        # test_wave_fp32_vs_float_typed_oracle between two fp32 families
        fa, ra, _, fa_all, ra_all = whole_and_blocks("auto")
        err = torch.maximum((fa_all - f).abs(), (ra_all - r).abs()).double().cpu().numpy()
        print(f"fp32 AUTO blocks vs LANE_FMA: median {np.median(err):.2e} p99 {np.quantile(err, 0.99):.2e} max {err.max():.2e}")
        assert np.median(err) <= 2e-5 and np.quantile(err, 0.99) <= 5e-3
        assert float(fa_all.abs().max()) <= amax * (1 + 1e-6) and float(ra_all.abs().max()) <= amax * (1 + 1e-6)


def test_config2_all_families_vs_oracle(torch_cuda, oracle):
    """BASELINE config 2 at its exact size -- batch 4096, N=10, fp64 -- every kernel family against the ORACLE:
    identical iteration counts; LANE bit for bit, WAVE (what AUTO runs here) and LANE_FMA within 1e-9."""
    from trajectory_controller_amd.synth import compact_inputs
    torch = torch_cuda
    v, dy, dphi = compact_inputs(10, 4096)
    of, orr, oit = oracle.solve_compact(10, v, dy, dphi, nthreads=8)
    tv, ty, tp = _dev(torch, v, dy, dphi)
    for algo in ("lane", "wave", "lane_fma", "auto"):
        with _solver(10, algo) as s:
            f, r, it = s.solve_batch_compact(tv, ty, tp, want_iters=True)
            f, r, it = f.cpu().numpy(), r.cpu().numpy(), it.cpu().numpy()
        assert np.array_equal(it, oit), algo
        if algo == "lane":
            assert bits_equal(f, of) and bits_equal(r, orr)
        else:
            assert max(np.abs(f - of).max(), np.abs(r - orr).max()) <= WAVE_ATOL, algo


@pytest.mark.parametrize("algo", ["lane", "wave"])
def test_fp32_tolerance_sweep(torch_cuda, algo):
    """BASELINE config 5 flavour: fp32 against the fp64 golden vectors.  fp32 cannot meet 1e-6
    (SURVEY.md fact 3); the histogram is printed, only gross sanity is asserted."""
    torch = torch_cuda
    for H in (5, 10, 20):
        g = load_golden(f"compact_H{H}.npz")
        v, dy, dphi = _dev(torch, g["v"], g["dy"], g["dphi"], dtype=torch.float32)
        with _solver(H, algo, dtype="f32") as s:
            f, r = s.solve_batch_compact(v, dy, dphi)
            f, r = f.cpu().numpy().astype(np.float64), r.cpu().numpy().astype(np.float64)
        err = np.maximum(np.abs(f - g["front"]), np.abs(r - g["rear"]))
        hist = {t: float(np.mean(err <= t)) for t in (1e-2, 1e-3, 1e-4, 1e-5, 1e-6)}
        print(f"fp32 {algo} H={H}: fraction within tol {hist}")
        assert np.all(np.isfinite(f)) and np.all(np.isfinite(r))
        assert np.median(err) < 1e-3
        assert hist[1e-2] > 0.5


@pytest.mark.parametrize("H,n", [(5, 4096), (10, 4096), (20, 2048)])
def test_wave_fp32_vs_float_typed_oracle(torch_cuda, oracle32, H, n):
    """WAVE in fp32 against the float-typed restatement (dlib's operation sequence typed float; dlib itself is
    fp64-only, so this pins nothing to the reference -- "parity unpinned" for fp32 -- but it is the one statement fp32
    admits): the WAVE kernels sum in a different order and use FMAs, so the comparison is a tolerance one -- the
    fraction of equal iteration counts and the error distribution, both bounded."""
    from trajectory_controller_amd.synth import compact_inputs
    torch = torch_cuda
    v, dy, dphi = (a.astype(np.float32) for a in compact_inputs(H, n, first=52000))
    of, orr, oit = oracle32.solve_compact(H, v, dy, dphi, nthreads=8)
    tv, ty, tp = _dev(torch, v, dy, dphi)
    with _solver(H, "wave", dtype="f32") as s:
        f, r, it = s.solve_batch_compact(tv, ty, tp, want_iters=True)
        f, r, it = f.cpu().numpy(), r.cpu().numpy(), it.cpu().numpy()
    same = float(np.mean(it == oit))
    err = np.maximum(np.abs(f - of), np.abs(r - orr))
    print(f"WAVE fp32 H={H}: equal iteration counts {same:.4f}; |du| median {np.median(err):.2e} p99 {np.quantile(err, 0.99):.2e} max {err.max():.2e}")
    floor = {5: 0.9, 10: 0.8, 20: 0.4}[H]
    assert same >= floor and np.median(err) <= 2e-5 and np.quantile(err, 0.99) <= 5e-3


def test_mixed_horizons_65536(torch_cuda, oracle):
    """BASELINE config 5: batch 65 536 split evenly over N in {5, 10, 20, 40}, fp64 parity against
    the golden vectors (each horizon's first 1024 instances are the fixture ones) and fp32 sweep."""
    from trajectory_controller_amd.synth import compact_inputs
    torch = torch_cuda
    Hs, per = (5, 10, 20, 40), 16384
    parts = [compact_inputs(H, per) for H in Hs]
    v, dy, dphi = (np.concatenate([p[c] for p in parts]) for c in range(3))
    hz = np.repeat(np.array(Hs), per)
    perm = np.random.default_rng(3).permutation(len(hz))          # interleave the horizons
    tv, ty, tp = _dev(torch, v[perm], dy[perm], dphi[perm])
    with _solver(20, "lane") as s:
        f, r, it = s.solve_batch_compact_mixed(hz[perm], tv, ty, tp, want_iters=True)
    with _solver(20, "auto") as s:   # 16 384 per horizon: AUTO picks WAVE for every bin (N = 40: the prefix-sum form); the four bins run concurrently
        fa, ra = s.solve_batch_compact_mixed(hz[perm], tv, ty, tp)
    assert float((fa - f).abs().max()) <= WAVE_ATOL and float((ra - r).abs().max()) <= WAVE_ATOL
    inv = np.argsort(perm)
    f, r, it = f.cpu().numpy()[inv], r.cpu().numpy()[inv], it.cpu().numpy()[inv]
    for b, H in enumerate(Hs):
        g = load_golden(f"compact_H{H}.npz")
        sl = slice(b * per, b * per + 1024)
        assert bits_equal(f[sl], g["front"]) and bits_equal(r[sl], g["rear"]), H
        assert it[b * per:(b + 1) * per].max() <= 10000
    with _solver(20, "auto", dtype="f32") as s:
        f32, r32 = s.solve_batch_compact_mixed(hz[perm], tv.float(), ty.float(), tp.float())
    f32, r32 = f32.cpu().numpy().astype(np.float64)[inv], r32.cpu().numpy().astype(np.float64)[inv]
    err = np.maximum(np.abs(f32 - f), np.abs(r32 - r))
    for b, H in enumerate(Hs):
        e = err[b * per:(b + 1) * per]
        print(f"fp32 vs fp64, H={H}: within " + ", ".join(f"{t:g}: {np.mean(e <= t):.3f}" for t in (1e-2, 1e-3, 1e-4, 1e-5, 1e-6)))
        assert np.median(e) < 1e-3


@pytest.mark.parametrize("algo", ["lane", "wave"])
def test_phase_boundaries_vs_oracle(torch_cuda, oracle, algo):
    """The coordinate-descent / projected-gradient hand-over (mpc.h:319-335): smo_iters and
    max_iter around each other, including 0 and caps inside the coordinate-descent phase."""
    from trajectory_controller_amd.synth import compact_inputs
    from trajectory_controller_amd import FLAG_MAX_ITER
    H, n = 10, 640
    v, dy, dphi = compact_inputs(H, n, first=7777)
    for smo, cap in ((50, 0), (50, 1), (50, 49), (50, 50), (50, 51), (50, 300), (0, 400), (1, 400),
                     (7, 10000), (200, 10000), (1000, 120)):
        of, orr, oit = oracle.solve_compact(H, v, dy, dphi, max_iter=cap, smo_iters=smo, nthreads=8)
        with _solver(H, algo, smo_iters=smo, max_iter=cap) as s:
            f, r, it = s.solve_batch_compact(v, dy, dphi, want_iters=True)
            flags = s.last_flags
        assert np.array_equal(it, oit), (smo, cap)
        if algo == "lane":
            assert bits_equal(f, of) and bits_equal(r, orr), (smo, cap)
        else:
            assert max(np.abs(f - of).max(), np.abs(r - orr).max()) <= WAVE_ATOL, (smo, cap)
        assert bool(flags & FLAG_MAX_ITER) == bool(np.any(oit >= cap)), (smo, cap)


def test_two_batches_in_flight(torch_cuda):
    """bench.py's steady state: two handles on two streams, batches alternating between them
    (tpc_mpc_reserve up front so that no solve allocates).  Each batch's outputs must be the
    single-batch outputs bit for bit, whatever the interleaving of the kernels was."""
    from trajectory_controller_amd.synth import compact_inputs
    torch = torch_cuda
    H, n = 20, 98304
    sets = [_dev(torch, *compact_inputs(H, n, first=7 + 1000 * b)) for b in range(4)]
    with _solver(H, "lane") as ref:
        want = [ref.solve_batch_compact(*x, want_iters=True) for x in sets]
    torch.cuda.synchronize()
    solvers = [_solver(H, "lane"), _solver(H, "lane")]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    for sv in solvers:
        sv.reserve(n)
    got = [None] * 8
    for k in range(8):
        with torch.cuda.stream(streams[k % 2]):
            got[k] = solvers[k % 2].solve_batch_compact(*sets[k % 4], want_iters=True, want_flags=False)
    torch.cuda.synchronize()
    for k in range(8):
        f, r, it = got[k]
        wf, wr, wit = want[k % 4]
        assert torch.equal(it, wit)
        assert bits_equal(f.cpu().numpy(), wf.cpu().numpy()) and bits_equal(r.cpu().numpy(), wr.cpu().numpy())
    for sv in solvers:
        sv.close()


def test_work_hint_changes_order_not_results(torch_cuda):
    """tpc_mpc_set_work_hint: the previous cycle's iteration counts as queue order.  Outputs and
    iteration counts must be bit-identical with no hint, a perfect hint, a reversed (shortest-first)
    hint and a host-memory hint; the perfect hint must need fewer wave-iterations than lambda."""
    from trajectory_controller_amd.synth import compact_inputs
    torch = torch_cuda
    H, n = 20, 131072
    v, dy, dphi = _dev(torch, *compact_inputs(H, n, first=99))
    with _solver(H, "lane") as s:
        f0, r0, it0 = s.solve_batch_compact(v, dy, dphi, want_iters=True)
        w0, _ = s.last_lane_stats()
        s.set_work_hint(it0)
        f1, r1, it1 = s.solve_batch_compact(v, dy, dphi, want_iters=True)
        w1, _ = s.last_lane_stats()
        f2, r2, it2 = s.solve_batch_compact(v, dy, dphi, want_iters=True)   # hint was one-shot
        w2, _ = s.last_lane_stats()
        s.set_work_hint((it0.max() + 1 - it0).contiguous())                  # shortest first: worst case
        f3, r3, it3 = s.solve_batch_compact(v, dy, dphi, want_iters=True)
        w3, _ = s.last_lane_stats()
        s.set_work_hint(it0.cpu().numpy())                                   # host copy
        f4, r4, it4 = s.solve_batch_compact(v, dy, dphi, want_iters=True)
        w4, _ = s.last_lane_stats()
        s.set_work_hint(it0[: n // 2].contiguous())                          # wrong size: ignored
        f5, r5, it5 = s.solve_batch_compact(v, dy, dphi, want_iters=True)
    for f, r, it in ((f1, r1, it1), (f2, r2, it2), (f3, r3, it3), (f4, r4, it4), (f5, r5, it5)):
        assert torch.equal(it, it0)
        assert bits_equal(f.cpu().numpy(), f0.cpu().numpy()) and bits_equal(r.cpu().numpy(), r0.cpu().numpy())
    print(f"wave-iterations: lambda order {w0}, perfect hint {w1}, no hint again {w2}, reversed hint {w3}, host hint {w4}")
    assert w1 < 0.97 * w0 and w4 < 0.97 * w0
    assert abs(w2 - w0) < 0.02 * w0


@pytest.mark.parametrize("H", [10, 20])
def test_lane_stop_test_builds_vs_oracle(torch_cuda, oracle, H):
    """The fused projected-gradient kernel exists in two builds (mpc_lane.h): a select-free stop test
    for batches every instance of which passes CompactModel::fast_stop_ok, and dlib's compare-and-
    select form for the rest.  Both must reproduce dlib bit for bit: bounds straddling zero with most
    variables pinned (fast build), bounds that do not straddle zero, an absurd speed in an otherwise
    ordinary batch, and an eps beyond the screen (all exact build)."""
    from trajectory_controller_amd.synth import compact_inputs
    n = 4096
    v, dy, dphi = compact_inputs(H, n, first=424242)
    cases = [
        dict(lo=(-0.05, -0.02), hi=(0.03, 0.05)),       # fast build, saturating almost everywhere
        dict(lo=(-0.1, -0.384), hi=(0.384, 0.2)),       # fast build, asymmetric
        dict(lo=(0.0, 0.0), hi=(0.3, 0.3)),             # start point ON the lower bound: exact build
        dict(lo=(0.05, -0.3), hi=(0.3, -0.05)),         # start point outside the bounds: exact build
        dict(lo=(-0.384, -0.384), hi=(0.384, 0.384), eps=2e30),
    ]
    for c in cases:
        eps = c.get("eps", 0.01)
        of, orr, oit = oracle.solve_compact(H, v, dy, dphi, lo=c["lo"], hi=c["hi"], eps=eps, nthreads=8)
        with _solver(H, "lane", lower=c["lo"], upper=c["hi"], eps=eps) as s:
            f, r, it = s.solve_batch_compact(v, dy, dphi, want_iters=True)
        assert np.array_equal(it, oit), c
        assert bits_equal(f, of) and bits_equal(r, orr), c
    # one absurd instance (|T*v| > 1e60) sends the whole batch through the exact build
    v2 = v.copy()
    v2[17] = 3e62
    of, orr, oit = oracle.solve_compact(H, v2, dy, dphi, nthreads=8)
    with _solver(H, "lane") as s:
        f, r, it = s.solve_batch_compact(v2, dy, dphi, want_iters=True)
    assert np.array_equal(it, oit)
    assert bits_equal(f, of) and bits_equal(r, orr)


@pytest.mark.parametrize("algo", ["lane", "wave"])
def test_zero_qdiag_continue_branch(torch_cuda, oracle, algo):
    """Q = (2, 0) as in dlib's own test makes Q_diag[H-1] = 0: the `continue` of mpc.h:322 (an
    iteration that counts but updates nothing, and may skip the v := u of mpc.h:330-334)."""
    rng = np.random.default_rng(21)
    I, H, n = 1, 10, 500
    A = np.tile(np.array([1.0, 1.0, 0.0, 1.0]), (n, 1)) + rng.uniform(-0.05, 0.05, (n, 4)) * np.array([0, 1, 0, 0])
    B = np.tile(np.array([0.0, 1.0]), (n, 1))
    Cc = rng.uniform(-0.05, 0.1, (n, 2))
    Q = np.tile(np.array([2.0, 0.0]), (n, 1))
    R = rng.uniform(0.5, 2.0, (n, 1))
    lo, hi = np.full((n, 1), -0.2), np.full((n, 1), 0.2)
    x0 = rng.uniform(-5, 5, (n, 2)) * np.array([1.0, 0.2])
    tg = np.zeros((n, H, 2))
    for smo in (50, 3):
        u0, cout, it = oracle.solve_general(I, H, A, B, Cc, Q, R, lo, hi, x0, tg, smo_iters=smo, nthreads=8)
        with _solver(H, algo, smo_iters=smo) as s:
            gu0, git = s.solve_batch_general(*[_soa(a) for a in (A, B, Cc, Q, R, lo, hi, x0, tg)], inputs=I,
                                             want_iters=True)
        assert np.array_equal(git, it), smo
        if algo == "lane":
            assert bits_equal(gu0.T, u0), smo
        else:
            assert np.abs(gu0.T - u0).max() <= WAVE_ATOL, smo


@pytest.mark.parametrize("algo", ["lane", "wave"])
def test_general_invalid_models_are_flagged(torch_cuda, oracle, algo):
    """Per-instance models that break dlib's requires clause (mpc_abstract.h:90-97) are not solved:
    start point back, iteration 0, TPC_MPC_FLAG_BAD_MODEL; their neighbours are unaffected."""
    from trajectory_controller_amd import FLAG_BAD_MODEL
    from trajectory_controller_amd.synth import general_inputs
    I, H, n = 2, 10, 256
    g = general_inputs(H, n, I=I, first=99)
    u0, _, it = oracle.solve_general(I, H, g["A"], g["B"], g["C"], g["Q"], g["R"], g["lo"], g["hi"], g["x0"],
                                     g["targets"], nthreads=4)
    bad = {3: ("R", 0, 0.0), 77: ("Q", 1, -1.0), 200: ("hi", 0, -1.0)}   # R = 0, Q < 0, hi < lo
    for k, (name, c, val) in bad.items():
        g[name][k, c] = val
    names = ["A", "B", "C", "Q", "R", "lo", "hi", "x0", "targets"]
    with _solver(H, algo) as s:
        gu0, git = s.solve_batch_general(*[_soa(g[k]) for k in names], inputs=I, want_iters=True)
        flags = s.last_flags
    assert flags & FLAG_BAD_MODEL
    good = np.ones(n, dtype=bool)
    good[list(bad)] = False
    assert np.all(gu0.T[~good] == 0) and np.all(git[~good] == 0)
    assert np.array_equal(git[good], it[good])
    if algo == "lane":
        assert bits_equal(gu0.T[good], u0[good])
    else:
        assert np.abs(gu0.T[good] - u0[good]).max() <= WAVE_ATOL


# ---------------------------------------------------------------------------------------------
# every shipped kernel instantiation under an oracle / real-dlib check (round 2)

ALL_H = [4, 5, 10, 20, 30, 40]
GEN_NAMES = ["A", "B", "C", "Q", "R", "lo", "hi", "x0", "targets"]


@pytest.mark.parametrize("I", [1, 2])
@pytest.mark.parametrize("H", ALL_H)
@pytest.mark.parametrize("algo", ["lane", "wave"])
def test_rollout8_golden(torch_cuda, I, H, algo):
    """8 controllers x 5 warm-started closed-loop steps per horizon, expected values from real dlib
    (tests/golden/make_golden_r02.py).  tpc_mpc_rollout keeps controls and dlib's v on the device
    between steps, i.e. this is the state-returning lane_pg_kernel at every horizon."""
    g = load_golden(f"rollout8_I{I}_H{H}.npz")
    steps, n = int(g["steps"]), g["A"].shape[0]
    with _solver(H, algo) as s:
        c, st, _ = s.rollout(steps, *[_soa(g[k]) for k in GEN_NAMES],
                             new_last_targets=_soa(g["new_last_targets"]), inputs=I)
    c = c.T.reshape(n, steps, I)
    st = st.T.reshape(n, steps, 2)
    if algo == "lane":
        assert bits_equal(c, g["controls"]) and bits_equal(st, g["states"])
    else:
        assert np.abs(c - g["controls"]).max() <= 1e-7


@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("I", [1, 2])
@pytest.mark.parametrize("H", ALL_H)
def test_lane_state_returning_vs_oracle(torch_cuda, oracle, oracle32, H, I, dtype):
    """controls_inout + v_inout: the caller gets the whole controller state back, which takes the
    unfused lane_pg_kernel (every horizon, both input counts, both dtypes).  Warm start from random
    controls and a random v; outputs, solved sequence, v and iteration counts against the oracle of
    the same arithmetic type, bit for bit."""
    from trajectory_controller_amd.synth import general_inputs
    orc, npdt, eq = (oracle, np.float64, bits_equal) if dtype == "f64" else (oracle32, np.float32, bits_equal32)
    n = 320 if H <= 20 else 130   # not multiples of 64
    g = {k: a.astype(npdt) for k, a in general_inputs(H, n, I=I, first=1234 + H).items()}
    rng = np.random.default_rng(100 + H + I)
    cin = rng.uniform(-0.3, 0.3, size=(n, H, I)).astype(npdt)
    vin = rng.uniform(-0.3, 0.3, size=(n, H, I)).astype(npdt)
    u0, cout, it, vout = orc.solve_general(I, H, *[g[k] for k in GEN_NAMES], controls_in=cin, v_in=vin,
                                           want_v=True, nthreads=8)
    controls, vstate = _soa(cin), _soa(vin)
    with _solver(H, "lane", dtype=dtype) as s:
        gu0, git = s.solve_batch_general(*[_soa(g[k]) for k in GEN_NAMES], controls=controls, v_state=vstate,
                                         inputs=I, want_iters=True)
    assert np.array_equal(git, it)
    assert eq(gu0.T, u0)
    assert eq(controls.T.reshape(n, H, I), cout)
    assert eq(vstate.T.reshape(n, H, I), vout)


@pytest.mark.parametrize("H", ALL_H)
def test_fp32_lane_compact_bit_exact(torch_cuda, oracle32, H):
    """The fp32 fused LANE kernels (compact form, both stop-test builds) against the float-typed
    restatement: same bits, same iteration counts."""
    from trajectory_controller_amd.synth import compact_inputs
    n = 3000 if H <= 20 else 700
    v, dy, dphi = (a.astype(np.float32) for a in compact_inputs(H, n, first=250000))
    cases = [dict(), dict(lo=(0.0, -0.3), hi=(0.3, 0.0))]      # second: bounds touch zero -> exact stop test
    for c in cases:
        kw = dict(lo=c["lo"], hi=c["hi"]) if c else {}
        of, orr, oit = oracle32.solve_compact(H, v, dy, dphi, nthreads=8, **kw)
        over = dict(lower=c["lo"], upper=c["hi"]) if c else {}
        with _solver(H, "lane", dtype="f32", **over) as s:
            f, r, it = s.solve_batch_compact(v, dy, dphi, want_iters=True)
        assert f.dtype == np.float32
        assert np.array_equal(it, oit), c
        assert bits_equal32(f, of) and bits_equal32(r, orr), c


@pytest.mark.parametrize("I", [1, 2])
@pytest.mark.parametrize("H", ALL_H)
def test_fp32_lane_general_bit_exact(torch_cuda, oracle32, H, I):
    """The fp32 fused LANE kernels, general form (cold start), against the float-typed restatement."""
    from trajectory_controller_amd.synth import general_inputs
    n = 700 if H <= 20 else 200
    g = {k: a.astype(np.float32) for k, a in general_inputs(H, n, I=I, first=55 + H).items()}
    u0, _, it = oracle32.solve_general(I, H, *[g[k] for k in GEN_NAMES], nthreads=8)
    with _solver(H, "lane", dtype="f32") as s:
        gu0, git = s.solve_batch_general(*[_soa(g[k]) for k in GEN_NAMES], inputs=I, want_iters=True)
    assert np.array_equal(git, it)
    assert bits_equal32(gu0.T, u0)


def test_signed_zero_outputs(torch_cuda, oracle):
    """bits_equal distinguishes +0 from -0.  The compact model drops products with the literal 0
    entries of A and B (mpc_model.h), which is value-exact but could flip the sign of a zero
    intermediate; this batch is built to produce exact zeros (zero targets, targets that cancel,
    one variable pinned at a bound that is 0) and must still match the oracle bit for bit."""
    H = 10
    v = np.array([1.0, 2.0, 0.5, 1.0, 3.0, 1.0, 1.0, 2.5])
    dy = np.array([0.0, -0.0, 0.0, 0.1, -0.1, 0.0, -0.0, 0.3])
    dphi = np.array([0.0, 0.0, -0.0, 0.0, -0.0, 0.2, -0.2, 0.0])
    for lo, hi in (((-0.384, -0.384), (0.384, 0.384)), ((0.0, -0.384), (0.384, 0.0)), ((-0.384, 0.0), (0.0, 0.384))):
        of, orr, oit = oracle.solve_compact(H, v, dy, dphi, lo=lo, hi=hi)
        with _solver(H, "lane", lower=lo, upper=hi) as s:
            f, r, it = s.solve_batch_compact(v, dy, dphi, want_iters=True)
        assert np.array_equal(it, oit), (lo, hi)
        assert bits_equal(f, of) and bits_equal(r, orr), (lo, hi, f, of, r, orr)


@pytest.mark.parametrize("I", [1, 2])
def test_general_stop_test_builds_vs_oracle(torch_cuda, oracle, I):
    """The general-form fused kernel has the two stop-test builds too (fp64): the "moved" form when
    GeneralModel::fast_stop_ok proves per instance that nothing can overflow and the bounds straddle
    zero, dlib's compare-and-select form otherwise.  Ordinary batch (fast build); one instance whose
    bound touches zero, one whose A would overflow the magnitude screen, one with a huge eps (each
    sends its batch through the exact build).  All against the oracle, bits and iteration counts."""
    from trajectory_controller_amd.synth import general_inputs
    H, n = 20, 1500
    base = general_inputs(H, n, I=I, first=8800)

    def run(g, **kw):
        u0, _, it = oracle.solve_general(I, H, *[g[k] for k in GEN_NAMES], nthreads=8, **kw)
        with _solver(H, "lane", **kw) as s:
            gu0, git = s.solve_batch_general(*[_soa(g[k]) for k in GEN_NAMES], inputs=I, want_iters=True)
        assert np.array_equal(git, it)
        assert bits_equal(gu0.T, u0)
    run(base)
    g = {k: a.copy() for k, a in base.items()}
    g["lo"][7, 0] = 0.0                                   # start point ON the bound
    run(g)
    g = {k: a.copy() for k, a in base.items()}
    g["A"][11] = [1.0, 3e7, 0.0, 1.0]                     # |A|^(2H) ~ 1e300: beyond the screen, still finite in dlib
    run(g)
    run(base, eps=2e30)
    g = {k: a.copy() for k, a in base.items()}
    g["targets"][5, 3, 1] = np.nan                        # a NaN in the linear term: dlib ignores NaN gradients
    g["targets"][9, 0, 0] = np.inf
    g["x0"][13, 0] = np.nan
    run(g)


@pytest.mark.parametrize("I", [1, 2])
def test_wave_mask_forms_vs_oracle(torch_cuda, oracle, I):
    """The WAVE kernel's two forms of the masked |df| (arithmetic where the model's screen allows it,
    dlib's compares otherwise): an ordinary batch, a warm start outside the bounds, a bound that
    touches zero and NaN / inf targets -- identical iteration counts, outputs within the family's
    tolerance (NaN where dlib gives NaN)."""
    from trajectory_controller_amd.synth import general_inputs
    H, n = 10, 400
    base = general_inputs(H, n, I=I, first=4400)
    rng = np.random.default_rng(17)

    def run(g, cin=None):
        u0, cout, it = oracle.solve_general(I, H, *[g[k] for k in GEN_NAMES], controls_in=cin, nthreads=8)
        controls = None if cin is None else _soa(cin)
        with _solver(H, "wave") as s:
            gu0, git = s.solve_batch_general(*[_soa(g[k]) for k in GEN_NAMES], controls=controls, inputs=I, want_iters=True)
        assert np.array_equal(git, it)
        assert np.array_equal(np.isnan(gu0.T), np.isnan(u0))
        assert np.nanmax(np.abs(gu0.T - u0), initial=0.0) <= WAVE_ATOL
    run(base)
    run(base, cin=rng.uniform(-0.6, 0.6, size=(n, H, I)))   # many start points outside +-0.384
    g = {k: a.copy() for k, a in base.items()}
    g["lo"][7, 0] = 0.0
    g["targets"][5, 3, 1] = np.nan
    g["targets"][9, 0, 0] = np.inf
    run(g)


@pytest.mark.parametrize("H", [4, 5, 10, 20, 30, 40])
def test_wave_queue_vs_oracle(torch_cuda, oracle, H):
    """More instances than one wavefront per SIMD (fp64, up to 32 variables: two or four instances per wavefront)
    or than the WAVE family's persistent grid holds (two wavefronts per SIMD): they are taken from a
    longest-first queue, ordered by lambda or by the caller's work hint.  The order must not change a
    result: iteration counts equal the oracle's, outputs within the family's tolerance, with and without a
    hint, compact and general form, and a batch that fits the grid (no queue) agrees on its share."""
    from trajectory_controller_amd.synth import compact_inputs, general_inputs
    n = 5000 if H < 30 else (3000 if H == 30 else 1500)   # (the 80-variable kernel's grid holds 1 024)
    v, dy, dphi = compact_inputs(H, n, first=31000 + H)
    of, orr, oit = oracle.solve_compact(H, v, dy, dphi, nthreads=8)
    with _solver(H, "wave") as s:
        f, r, it = s.solve_batch_compact(v, dy, dphi, want_iters=True)
        assert np.array_equal(it, oit)
        assert max(np.abs(f - of).max(), np.abs(r - orr).max()) <= WAVE_ATOL
        s.set_work_hint(np.ascontiguousarray(oit[::-1].astype(np.int32)))    # a deliberately bad order
        f2, r2, it2 = s.solve_batch_compact(v, dy, dphi, want_iters=True)
        assert np.array_equal(it2, oit) and np.array_equal(f2, f) and np.array_equal(r2, r)
        s.set_work_hint(None)
        f3, r3, it3 = s.solve_batch_compact(v[:700], dy[:700], dphi[:700], want_iters=True)   # fits the grid
        assert np.array_equal(it3, oit[:700]) and np.array_equal(f3, f[:700]) and np.array_equal(r3, r[:700])
        for I in (1, 2):
            if I * H > 64 and I != 2:
                continue
            g = general_inputs(H, 2600 if H < 40 else 1300, I=I, first=52000 + H)
            u0, _, git0 = oracle.solve_general(I, H, *[g[k] for k in GEN_NAMES], nthreads=8)
            gu0, git = s.solve_batch_general(*[_soa(g[k]) for k in GEN_NAMES], inputs=I, want_iters=True)
            assert np.array_equal(git, git0), I
            assert np.abs(gu0.T - u0).max() <= WAVE_ATOL, I
            if H == 10:   # a warm start (controls in and out) through the grouped path: four (I = 1) and two (I = 2) per wavefront
                cin = np.random.default_rng(5 + I).uniform(-0.5, 0.5, size=(g["A"].shape[0], H, I))
                wu0, wc, wit = oracle.solve_general(I, H, *[g[k] for k in GEN_NAMES], controls_in=cin, nthreads=8)
                controls = _soa(cin)
                gu0, git = s.solve_batch_general(*[_soa(g[k]) for k in GEN_NAMES], controls=controls, inputs=I, want_iters=True)
                assert np.array_equal(git, wit), I
                assert np.abs(gu0.T - wu0).max() <= WAVE_ATOL, I
                assert np.abs(controls.T.reshape(wc.shape) - wc).max() <= WAVE_ATOL, I


@pytest.mark.parametrize("H", [10, 20])
def test_wave_queue_fp32_equals_unqueued(torch_cuda, H):
    """fp32 WAVE has no oracle of its own (it is a tolerance-sweep mode), but its work queue -- one instance per
    wavefront, static rounds plus ticket counters -- must not change a bit: a batch large enough for the dynamic
    part of the queue equals the same instances solved in chunks that fit the chip at once (no queue)."""
    from trajectory_controller_amd.synth import compact_inputs
    n = 5000
    v, dy, dphi = (a.astype(np.float32) for a in compact_inputs(H, n, first=64000 + H))
    with _solver(H, "wave", dtype="f32") as s:
        f, r, it = s.solve_batch_compact(v, dy, dphi, want_iters=True)
        for a in range(0, n, 1000):
            fc, rc, ic = s.solve_batch_compact(v[a:a + 1000], dy[a:a + 1000], dphi[a:a + 1000], want_iters=True)
            assert np.array_equal(ic, it[a:a + 1000]), a
            assert np.array_equal(fc.view(np.uint32), f[a:a + 1000].view(np.uint32)), a
            assert np.array_equal(rc.view(np.uint32), r[a:a + 1000].view(np.uint32)), a


@pytest.mark.parametrize("H", [4, 5, 10])
def test_wave_groups_equal_one_per_wavefront(torch_cuda, H):
    """fp64, up to 32 variables: two (N = 10) or four (N = 4, 5) instances per wavefront run every instance through
    the arithmetic of the one-instance kernel (maxima are exact, the sums associate alike), so a grouped batch
    equals the same instances solved in chunks of one per wavefront BIT FOR BIT -- iteration counts and outputs."""
    from trajectory_controller_amd.synth import compact_inputs
    n = 9000
    v, dy, dphi = compact_inputs(H, n, first=83000 + H)
    with _solver(H, "wave") as s:
        f, r, it = s.solve_batch_compact(v, dy, dphi, want_iters=True)
        for a in range(0, n, 1000):
            fc, rc, ic = s.solve_batch_compact(v[a:a + 1000], dy[a:a + 1000], dphi[a:a + 1000], want_iters=True)
            assert np.array_equal(ic, it[a:a + 1000]), a
            assert bits_equal(fc, f[a:a + 1000]) and bits_equal(rc, r[a:a + 1000]), a


def test_wave_one_instance_per_wavefront_switch(torch_cuda, oracle):
    """tpc_mpc_set_option(TPC_MPC_OPT_WAVE_GROUP, 1) keeps the fp64 WAVE batches of up to 32 variables on the
    one-instance-per-wavefront kernels the grouped path replaced by default: same iteration counts, same
    tolerance; 2 and 4 force pairs / fours."""
    from trajectory_controller_amd import capi
    from trajectory_controller_amd.synth import compact_inputs
    for H, n in ((10, 5000), (5, 3000)):
        v, dy, dphi = compact_inputs(H, n, first=91000 + H)
        of, orr, oit = oracle.solve_compact(H, v, dy, dphi, nthreads=8)
        for group in (1, 2, 4, 0):
            with _solver(H, "wave") as s:
                s.set_option(capi.OPT_WAVE_GROUP, group)
                f, r, it = s.solve_batch_compact(v, dy, dphi, want_iters=True)
            assert np.array_equal(it, oit), (H, group)
            assert max(np.abs(f - of).max(), np.abs(r - orr).max()) <= 1e-9, (H, group)


@pytest.mark.parametrize("H", [10, 5])
def test_wave_queue_boundaries(torch_cuda, oracle, H):
    """Batch sizes at every seam of the WAVE work queue (W persistent wavefronts = 8 per CU): where one instance
    per wavefront gives way to two (N = 10) and to four (N = 5), the end of the two static rounds, the first
    dynamic positions and the seams between the 16 ticket counters they are dealt from -- in units of one, two
    and four instances -- ragged last groups, and the largest batch the queue takes.  Every instance must be
    solved exactly once: iteration counts equal the oracle's and outputs within the family's tolerance."""
    import torch
    from trajectory_controller_amd.synth import compact_inputs
    W = 8 * torch.cuda.get_device_properties(0).multi_processor_count
    sizes = set()
    for unit in (1, 2, 4):          # instances per queue position
        for pos in (W // 2, W, 2 * W, 2 * W + 16):
            for d in (-1, 0, 1, 2, 3, 15 * unit, 16 * unit, 16 * unit + 1):
                sizes.add(unit * pos + d)
    sizes = sorted(n for n in sizes | {32767, 32768} if 0 < n <= 32768)
    v, dy, dphi = compact_inputs(H, max(sizes), first=77000 + H)
    of, orr, oit = oracle.solve_compact(H, v, dy, dphi, nthreads=8)
    with _solver(H, "wave") as s:
        for n in sizes:
            f, r, it = s.solve_batch_compact(v[:n], dy[:n], dphi[:n], want_iters=True)
            assert np.array_equal(it, oit[:n]), n
            assert max(np.abs(f - of[:n]).max(), np.abs(r - orr[:n]).max()) <= WAVE_ATOL, n


@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("H", [1, 2, 3, 7, 8, 15, 25, 33, 64])
def test_generic_horizon_vs_oracle(torch_cuda, oracle, oracle32, H, dtype):
    """Horizons without a specialised kernel (dlib's horizon is a template parameter: any value
    compiles there) run the generic kernel: compact form, general form cold, and general form with
    the controller state in and out -- bits and iteration counts against the oracle of the same type."""
    from trajectory_controller_amd.synth import compact_inputs, general_inputs
    orc, npdt, eq = (oracle, np.float64, bits_equal) if dtype == "f64" else (oracle32, np.float32, bits_equal32)
    n = 500 if H <= 33 else 130
    v, dy, dphi = (a.astype(npdt) for a in compact_inputs(H, n, first=77000))
    of, orr, oit = orc.solve_compact(H, v, dy, dphi, nthreads=8)
    with _solver(H, "auto", dtype=dtype) as s:
        f, r, it = s.solve_batch_compact(v, dy, dphi, want_iters=True)
        assert np.array_equal(it, oit)
        assert eq(f, of) and eq(r, orr)
        f1, r1 = s.solve_one(float(v[3]), float(dy[3]), float(dphi[3]))      # a launch of the generic kernel
        assert f1 == of[3] and r1 == orr[3]
        for I in (1, 2):
            g = {k: a.astype(npdt) for k, a in general_inputs(H, n, I=I, first=9100 + H).items()}
            u0, _, git0 = orc.solve_general(I, H, *[g[k] for k in GEN_NAMES], nthreads=8)
            gu0, git = s.solve_batch_general(*[_soa(g[k]) for k in GEN_NAMES], inputs=I, want_iters=True)
            assert np.array_equal(git, git0) and eq(gu0.T, u0), I
            rng = np.random.default_rng(H + I)
            cin = rng.uniform(-0.3, 0.3, size=(n, H, I)).astype(npdt)
            vin = rng.uniform(-0.3, 0.3, size=(n, H, I)).astype(npdt)
            u0, cout, git0, vout = orc.solve_general(I, H, *[g[k] for k in GEN_NAMES], controls_in=cin, v_in=vin,
                                                     want_v=True, nthreads=8)
            controls, vstate = _soa(cin), _soa(vin)
            gu0, git = s.solve_batch_general(*[_soa(g[k]) for k in GEN_NAMES], controls=controls, v_state=vstate,
                                             inputs=I, want_iters=True)
            assert np.array_equal(git, git0) and eq(gu0.T, u0), I
            assert eq(controls.T.reshape(n, H, I), cout) and eq(vstate.T.reshape(n, H, I), vout), I
