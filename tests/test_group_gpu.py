"""`-m gpu` tests of the GROUP kernel family (csrc/mpc_group.h: G lanes per instance), through the C ABI.

The family computes the LANE_FMA arithmetic in a third association (chunks of the horizon joined by scans over the
lanes of a group), so its statement is the tolerance families': against the reference -- the real-dlib golden vectors
and the pinned oracle -- |du| <= 1e-9 absolute and IDENTICAL iteration counts in fp64 (observed <= 3e-12); a control
dlib leaves on a bound is on it bit for bit.  fp32 has no reference to be held to (dlib is fp64-only): it is compared
with the float-typed restatement as a tolerance statement.  Every group size built for a horizon is covered, with
batch sizes that leave ragged groups and ragged wavefronts.
"""
import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu

GROUP_ATOL = 1e-9
GROUP = 4          # tpc_mpc_algo
LANE_FMA = 3
BUILT = [(10, 2), (10, 4), (20, 2), (20, 4), (20, 8), (30, 2), (30, 4), (30, 8), (40, 2), (40, 4), (40, 8)]


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch


def _solver(H, G=0, algo="group", dtype="f64", **kw):
    from trajectory_controller_amd import MpcSolver, capi
    s = MpcSolver(horizon=H, device=0, dtype=dtype, algo=algo, **kw)
    if G:
        s.set_option(capi.OPT_GROUP_LANES, G)
    return s


def _run(torch, s, v, dy, dphi, dtype=None, expect=GROUP):
    tv, ty, tp = (torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0", dtype=dtype) for a in (v, dy, dphi))
    s.set_profiling(True)
    f, r, it = s.solve_batch_compact(tv, ty, tp, want_iters=True)
    torch.cuda.synchronize()
    if expect is not None:
        assert s.last_kernel_times()[2] == expect      # the family under test ran
    return f.cpu().numpy(), r.cpu().numpy(), it.cpu().numpy()


@pytest.mark.parametrize("H,G", [(h, g) for h, g in BUILT if h != 30])
def test_group_golden(torch_cuda, H, G):
    """Real-dlib golden vectors: <= 1e-9, and every control dlib leaves on a bound is on it bit for bit."""
    g = load_golden(f"compact_H{H}.npz")
    with _solver(H, G) as s:
        f, r, it = _run(torch_cuda, s, g["v"], g["dy"], g["dphi"])
        assert s.last_flags & 1 == 0
    assert max(np.abs(f - g["front"]).max(), np.abs(r - g["rear"]).max()) <= GROUP_ATOL
    A = 22.0 * np.pi / 180.0
    assert np.array_equal(np.abs(g["front"]) == A, np.abs(f) == A)
    assert np.array_equal(np.abs(g["rear"]) == A, np.abs(r) == A)
    known = g["iters_lb"] >= 0
    assert np.all(it[known] >= g["iters_lb"][known])


@pytest.mark.parametrize("H,G", BUILT)
def test_group_vs_oracle_iters(torch_cuda, oracle, H, G):
    """Fresh seeded inputs against the pinned oracle, every group size built: identical iteration counts, |du| <= 1e-9.
    The batch leaves a ragged last wavefront and more instances than the first refill pass hands out."""
    from trajectory_controller_amd.synth import compact_inputs
    n = {10: 4099, 20: 3001, 30: 1203, 40: 701}[H]
    v, dy, dphi = compact_inputs(H, n, first=610000 + 100 * H + G)
    of, orr, oit = oracle.solve_compact(H, v, dy, dphi, nthreads=8)
    with _solver(H, G) as s:
        f, r, it = _run(torch_cuda, s, v, dy, dphi)
        f2, r2, it2 = _run(torch_cuda, s, v[::-1], dy[::-1], dphi[::-1])
    assert np.array_equal(it, oit)
    assert max(np.abs(f - of).max(), np.abs(r - orr).max()) <= GROUP_ATOL
    # an instance's result does not depend on its place in the batch or on the group that picks it up
    assert np.array_equal(f2[::-1], f) and np.array_equal(r2[::-1], r) and np.array_equal(it2[::-1], it)


@pytest.mark.parametrize("H,G", [(10, 2), (20, 4), (30, 8), (40, 8)])
@pytest.mark.parametrize("n", [1, 3, 17, 65])
def test_group_tiny_batches(torch_cuda, oracle, H, G, n):
    """Fewer instances than one wavefront carries, and one more than it carries."""
    from trajectory_controller_amd.synth import compact_inputs
    v, dy, dphi = compact_inputs(H, n, first=77 + n)
    of, orr, oit = oracle.solve_compact(H, v, dy, dphi)
    with _solver(H, G) as s:
        f, r, it = _run(torch_cuda, s, v, dy, dphi)
    assert np.array_equal(it, oit) and max(np.abs(f - of).max(), np.abs(r - orr).max()) <= GROUP_ATOL


@pytest.mark.parametrize("H,G", [(10, 4), (20, 4), (20, 8), (40, 8)])
@pytest.mark.parametrize("lo,hi,fast", [((-0.3, -0.2), (0.25, 0.4), True), ((0.05, -0.3), (0.3, -0.1), False),
                                         ((-1e-3, -0.5), (2e-3, 0.5), True)])
def test_group_other_bounds(torch_cuda, oracle, H, G, lo, hi, fast):
    """Unequal bounds (the build with one more addition per step); a box that does not contain the start point u = 0
    fails the fast stop test's screen: the batch then runs LANE_FMA's exact build on the same records."""
    from trajectory_controller_amd.synth import compact_inputs
    n = 900 if H <= 20 else 300
    v, dy, dphi = compact_inputs(H, n, first=7000)
    of, orr, oit = oracle.solve_compact(H, v, dy, dphi, lo=lo, hi=hi, nthreads=8)
    with _solver(H, G, lower=lo, upper=hi) as s:
        f, r, it = _run(torch_cuda, s, v, dy, dphi)
    assert np.array_equal(it, oit) and max(np.abs(f - of).max(), np.abs(r - orr).max()) <= GROUP_ATOL
    assert np.array_equal((of == lo[0]) | (of == hi[0]), (f == lo[0]) | (f == hi[0]))


@pytest.mark.parametrize("G", [2, 4])
@pytest.mark.parametrize("smo,cap", [(0, 10000), (1, 10000), (49, 10000), (50, 50), (50, 51), (7, 3), (200, 10000)])
def test_group_phase_boundaries(torch_cuda, oracle, G, smo, cap):
    """Coordinate-descent / projected-gradient hand-over and the iteration cap (mpc.h:271, :319, :330-334)."""
    from trajectory_controller_amd import FLAG_MAX_ITER
    from trajectory_controller_amd.synth import compact_inputs
    H, n = 10, 1000
    v, dy, dphi = compact_inputs(H, n, first=900)
    of, orr, oit = oracle.solve_compact(H, v, dy, dphi, smo_iters=smo, max_iter=cap, nthreads=8)
    with _solver(H, G, smo_iters=smo, max_iter=cap) as s:
        f, r, it = _run(torch_cuda, s, v, dy, dphi)
        assert bool(s.last_flags & FLAG_MAX_ITER) == bool((oit == cap).any())
    assert np.array_equal(it, oit) and max(np.abs(f - of).max(), np.abs(r - orr).max()) <= GROUP_ATOL


def test_group_knobs_and_edges(torch_cuda, oracle):
    from trajectory_controller_amd import FLAG_MAX_ITER, FLAG_NONFINITE
    from trajectory_controller_amd.synth import compact_inputs
    torch = torch_cuda
    g = load_golden("compact_knobs_H10.npz")
    with _solver(10, 2, eps=float(g["eps"]), max_iter=int(g["max_iter"])) as s:
        f, r, it = _run(torch, s, g["v"], g["dy"], g["dphi"])
    assert it.max() <= 300 and max(np.abs(f - g["front"]).max(), np.abs(r - g["rear"]).max()) <= GROUP_ATOL
    e = load_golden("compact_edge.npz")
    with _solver(20, 4) as s:
        f, r, it = s.solve_batch_compact(e["v"], e["dy"], e["dphi"], want_iters=True)   # host-memory path
        flags = s.last_flags
    assert np.nanmax(np.abs(f - e["front_H20"])) <= GROUP_ATOL and np.nanmax(np.abs(r - e["rear_H20"])) <= GROUP_ATOL
    assert np.all(f[:4] == 0) and np.all(r[:4] == 0) and np.all(it[:4] == 0) and flags & FLAG_NONFINITE
    assert flags & FLAG_MAX_ITER and it[11] == 10000           # the v = 50 row ends on the cap in dlib too
    # a tight eps: many projected-gradient iterations, ended by the cap for most
    v, dy, dphi = compact_inputs(20, 500, first=50)
    of, orr, oit = oracle.solve_compact(20, v, dy, dphi, eps=1e-6, max_iter=4000, nthreads=8)
    with _solver(20, 4, eps=1e-6, max_iter=4000) as s:
        f, r, it = _run(torch, s, v, dy, dphi)
    print(f"eps 1e-6: equal iteration counts {np.mean(it == oit):.3f}, max |du| {max(np.abs(f - of).max(), np.abs(r - orr).max()):.2e}")
    assert np.mean(it == oit) >= 0.99 and max(np.abs(f - of).max(), np.abs(r - orr).max()) <= 1e-7


@pytest.mark.parametrize("H,n", [(20, 16384), (40, 8192), (10, 32768), (30, 16384)])
def test_auto_takes_group_at_mid_size_batches(torch_cuda, oracle, H, n):
    """AUTO's middle: batches too large for a wavefront each and too small to fill the chip one lane each.  The family
    that ran is GROUP, and a sample of the batch is held to the oracle (identical iteration counts, <= 1e-9), the whole
    batch to LANE_FMA (identical iteration counts, <= 1e-9)."""
    from trajectory_controller_amd.synth import compact_inputs
    torch = torch_cuda
    v, dy, dphi = compact_inputs(H, n)
    with _solver(H, algo="auto") as s:
        f, r, it = _run(torch, s, v, dy, dphi, expect=GROUP)
    # (the whole batch against a one-lane family: LANE_FMA up to N = 20; at N = 30 / 40 in fp64 GROUP takes LANE_FMA's
    #  requests itself since round 5 -- asserted here -- and the partner is the bit-exact family)
    with _solver(H, algo="lane_fma") as s:
        lf, lr, lit = _run(torch, s, v, dy, dphi, expect=LANE_FMA if H < 30 else GROUP)
    if H >= 30:
        with _solver(H, algo="lane") as s:
            lf, lr, lit = _run(torch, s, v, dy, dphi, expect=2)
    assert np.array_equal(it, lit)
    assert max(np.abs(f - lf).max(), np.abs(r - lr).max()) <= GROUP_ATOL
    m = 1500 if H <= 20 else 500
    of, orr, oit = oracle.solve_compact(H, v[:m], dy[:m], dphi[:m], nthreads=8)
    assert np.array_equal(it[:m], oit) and max(np.abs(f[:m] - of).max(), np.abs(r[:m] - orr).max()) <= GROUP_ATOL


@pytest.mark.parametrize("dtype,H,n,expect", [
    ("f64", 40, 600000, GROUP),        # never overtaken at N = 30 / 40 (the table used to end at 524 288)
    ("f64", 30, 600000, GROUP),
    ("f64", 20, 200000, LANE_FMA),     # N = 20: the one-lane family from 160 530 on
    ("f32", 20, 262144, GROUP),        # fp32, two wavefronts per SIMD: up to 321 060
    ("f32", 20, 400000, LANE_FMA),
    ("f32", 40, 600000, GROUP),
])
def test_auto_at_the_top_of_the_table(torch_cuda, dtype, H, n, expect):
    """What AUTO runs past the sizes its crossovers were measured at (csrc/auto_table.h): a family that was still ahead
    at the top of the ladder stays in place where its cost per further instance is the lower one.  The iteration cap is
    cut to 60 so that the large batches take milliseconds (the choice of family does not depend on it), with the
    opt-out of AUTO's second pass for capped instances set; the outputs are then the named family's, bit for bit."""
    from trajectory_controller_amd import capi
    from trajectory_controller_amd.synth import compact_inputs
    torch = torch_cuda
    tdt = torch.float32 if dtype == "f32" else torch.float64
    tv, ty, tp = (torch.from_numpy(a).to("cuda:0", dtype=tdt) for a in compact_inputs(H, n, first=5_000_000))
    with _solver(H, algo="auto", dtype=dtype, max_iter=60, options=capi.PARAM_FAST_CAPPED) as s:
        s.set_profiling(True)
        f, r = s.solve_batch_compact(tv, ty, tp)
        torch.cuda.synchronize()
        assert s.last_kernel_times()[2] == expect
    with _solver(H, algo={GROUP: "group", LANE_FMA: "lane_fma"}[expect], dtype=dtype, max_iter=60, options=capi.PARAM_FAST_CAPPED) as s:
        f2, r2 = s.solve_batch_compact(tv, ty, tp)
        torch.cuda.synchronize()
    assert torch.equal(f, f2) and torch.equal(r, r2)


@pytest.mark.parametrize("H", [30, 40])
def test_auto_general_form_large_batch_stays_with_group(torch_cuda, H):
    """General form, N = 30 / 40: GROUP (G = 4) is never overtaken by the one-lane LANE kernels (393 216 x N = 40: 119 ms
    against 316) -- a batch beyond the old end of the table (262 144) must not drop to them."""
    from trajectory_controller_amd import capi
    from trajectory_controller_amd.synth import general_inputs
    torch = torch_cuda
    n = 300000
    gi = general_inputs(H, n, I=2)
    names = ["A", "B", "C", "Q", "R", "lo", "hi", "x0", "targets"]
    tv = [torch.from_numpy(np.ascontiguousarray(gi[k].reshape(n, -1).T)).to("cuda:0") for k in names]
    with _solver(H, algo="auto", max_iter=60, options=capi.PARAM_FAST_CAPPED) as s:
        s.set_profiling(True)
        u = s.solve_batch_general(*tv, inputs=2)
        torch.cuda.synchronize()
        assert s.last_kernel_times()[2] == GROUP
    assert bool(torch.isfinite(u).all())


@pytest.mark.parametrize("H,G,n", [(10, 2, 4096), (20, 4, 3000), (40, 8, 700)])
def test_group_fp32_vs_float_typed_oracle(torch_cuda, oracle32, H, G, n):
    """fp32 (unpinned: dlib is fp64-only): against the float-typed restatement as a tolerance statement, and against
    the fp32 LANE_FMA kernels, which do the same arithmetic in another association."""
    from trajectory_controller_amd.synth import compact_inputs
    torch = torch_cuda
    v, dy, dphi = (a.astype(np.float32) for a in compact_inputs(H, n, first=300000))
    of, orr, oit = oracle32.solve_compact(H, v, dy, dphi, nthreads=8)
    with _solver(H, G, dtype="f32") as s:
        f, r, it = _run(torch, s, v, dy, dphi)
    with _solver(H, algo="lane_fma", dtype="f32") as s:
        lf, lr, lit = _run(torch, s, v, dy, dphi, expect=LANE_FMA)
    err = np.maximum(np.abs(f - of), np.abs(r - orr))
    same, same_l = float(np.mean(it == oit)), float(np.mean(it == lit))
    print(f"fp32 GROUP H={H} G={G}: equal iteration counts vs restatement {same:.4f}, vs LANE_FMA {same_l:.4f}, "
          f"median |du| {np.median(err):.2e}, p99 {np.quantile(err, 0.99):.2e}")
    A = np.float32(22.0 * np.pi / 180.0)
    assert np.isfinite(f).all() and np.abs(f).max() <= A and np.abs(r).max() <= A and it.max() <= 10000
    assert same >= 0.4 and same_l >= 0.4 and np.median(err) <= 1e-4


@pytest.mark.parametrize("H,G,n,vscale", [(10, 2, 4096, 0.75), (10, 4, 2000, 0.75), (20, 2, 3000, 0.75), (20, 4, 3001, 0.75),
                                          (20, 8, 1500, 0.75), (30, 2, 1203, 0.3), (30, 4, 900, 0.3), (40, 4, 701, 0.25), (40, 8, 700, 0.25)])
def test_group_fp32_stop_test_builds_agree(torch_cuda, H, G, n, vscale):
    """fp32 GROUP has two builds of its kernel (group_pg_kernel's MOVED): the stop test read off the projected step where
    every instance of the batch passes ub::moved_stop_ok, dlib's mask as arithmetic otherwise -- LANE_FMA's MODE 2 / MODE 1.
    Where the screen holds the two take the same decisions on the same values: a batch inside the screen, and the same
    batch with ONE instance beyond it appended (which sends the whole batch to the mask build), give every shared
    instance the same bits and iteration counts.  Also at both grid sizes (one / two wavefronts per SIMD): a result does
    not depend on how many wavefronts share a SIMD."""
    from trajectory_controller_amd import capi
    from trajectory_controller_amd.synth import compact_inputs
    torch = torch_cuda
    v, dy, dphi = (a.astype(np.float32) for a in compact_inputs(H, n, first=410000 + H))
    # speeds inside the screen (lambda grows with v^2 and the horizon): <= 3 m/s at N = 10 / 20, <= 1.2 / 1 m/s at N = 30 / 40
    v = (v * np.float32(vscale)).astype(np.float32)
    v2, dy2, dphi2 = (np.concatenate([a, a[:1]]) for a in (v, dy, dphi))
    v2[-1] = np.float32(50.0)                                 # lambda ~ v^2: far beyond it (the edge fixture's fastest speed)
    with _solver(H, G, dtype="f32") as s:
        f, r, it = _run(torch, s, v, dy, dphi)
        fm, rm, itm = _run(torch, s, v2, dy2, dphi2)
        cus = torch.cuda.get_device_properties(0).multi_processor_count
        out = []
        for per_simd in (1, 2):
            s._check(s._lib.tpc_mpc_x_set_group_share(s._h, per_simd * 4 * cus, 0))
            out.append(_run(torch, s, v, dy, dphi))
    # (the batch inside the screen really ran the other build: LANE_FMA's own two builds are told apart the same way in
    # tests/test_ub_gpu.py::test_ub_f32_stop_test_builds -- here the statement is that nothing tells them apart)
    assert np.array_equal(it, itm[:n]) and np.array_equal(f, fm[:n]) and np.array_equal(r, rm[:n])
    for fo, ro, ito in out:
        assert np.array_equal(it, ito) and np.array_equal(f, fo) and np.array_equal(r, ro)
    A = np.float32(22.0 * np.pi / 180.0)
    assert np.isfinite(fm).all() and np.abs(fm).max() <= A and np.abs(rm).max() <= A


def test_group_falls_back_where_it_has_no_kernel(torch_cuda, oracle):
    """An explicit GROUP request at a horizon without group kernels (N = 4, 5), or with bounds the unit box cannot take
    (a pinned input), runs the one-lane families -- and says so through last_kernel_times."""
    from trajectory_controller_amd.synth import compact_inputs
    torch = torch_cuda
    v, dy, dphi = compact_inputs(5, 2000)
    of, orr, oit = oracle.solve_compact(5, v, dy, dphi, nthreads=8)
    with _solver(5) as s:
        f, r, it = _run(torch, s, v, dy, dphi, expect=LANE_FMA)
    assert np.array_equal(it, oit) and max(np.abs(f - of).max(), np.abs(r - orr).max()) <= GROUP_ATOL
    lo, hi = (-0.2, 0.1), (0.3, 0.1)                     # input 1 pinned: LANE (bit-exact)
    v, dy, dphi = compact_inputs(20, 500)
    of, orr, oit = oracle.solve_compact(20, v, dy, dphi, lo=lo, hi=hi, nthreads=8)
    with _solver(20, lower=lo, upper=hi) as s:
        f, r, it = _run(torch, s, v, dy, dphi, expect=2)
    assert np.array_equal(it, oit) and np.array_equal(f, of) and np.array_equal(r, orr)
