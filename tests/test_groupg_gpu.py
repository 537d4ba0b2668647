"""`-m gpu` tests of the GROUP kernels for the GENERAL model (csrc/mpc_groupg.h: G lanes per instance, per-instance
A, B, C, Q, R, bounds, x0, per-step targets), through the C ABI.

Tolerance-family statement against the reference (real-dlib golden vectors, the pinned oracle): |du| <= 1e-9 absolute
and IDENTICAL iteration counts in fp64, at every horizon with group kernels (N = 10, 20, 30, 40), one and two inputs,
every group size built, cold starts and warm starts with the controller state (controls and dlib's v) in and out, and
the closed loop of tpc_mpc_rollout.  A batch the stop-test screen refuses runs the one-lane families' exact kernels on
the same records.
"""
import numpy as np
import pytest

from conftest import bits_equal, load_golden

pytestmark = pytest.mark.gpu

ATOL = 1e-9
GROUP, LANE = 4, 2
GNAMES = ["A", "B", "C", "Q", "R", "lo", "hi", "x0", "targets"]
BUILT = [(10, 2), (10, 4), (20, 2), (20, 4), (20, 8), (30, 4), (30, 8), (40, 4), (40, 8)]


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch


def _soa(a):
    a = np.asarray(a)
    return np.ascontiguousarray(a.reshape(a.shape[0], -1).T)


def _solver(H, G=0, dtype="f64", **kw):
    from trajectory_controller_amd import MpcSolver, capi
    s = MpcSolver(horizon=H, device=0, dtype=dtype, algo="group", **kw)
    if G:
        s.set_option(capi.OPT_GROUP_LANES, G)
    return s


def _run(torch, s, g, I, dtype=None, expect=GROUP, **state):
    dev = [torch.from_numpy(_soa(g[k])).to("cuda:0", dtype=dtype) for k in GNAMES]
    s.set_profiling(True)
    u0, it = s.solve_batch_general(*dev, inputs=I, want_iters=True, **state)
    torch.cuda.synchronize()
    if expect is not None:
        assert s.last_kernel_times()[2] == expect
    return u0.cpu().numpy().T, it.cpu().numpy()


@pytest.mark.parametrize("I", [1, 2])
@pytest.mark.parametrize("H,G", BUILT)
def test_groupg_golden(torch_cuda, I, H, G):
    """Real-dlib golden vectors of the general form."""
    g = load_golden(f"general_I{I}_H{H}.npz")
    with _solver(H, G) as s:
        u0, it = _run(torch_cuda, s, g, I)
        assert s.last_flags & ~2 == 0        # (at N = 40 some fixture instances end on the iteration cap, in dlib too)
    assert np.abs(u0 - g["u0"]).max() <= ATOL


@pytest.mark.parametrize("I", [1, 2])
@pytest.mark.parametrize("H,G", BUILT)
def test_groupg_vs_oracle_iters(torch_cuda, oracle, I, H, G):
    """Fresh seeded inputs: identical iteration counts, |du| <= 1e-9, a control dlib leaves on a bound is on it bit
    for bit; ragged last wavefront, more instances than the first refill pass hands out."""
    from trajectory_controller_amd.synth import general_inputs
    n = {10: 4099, 20: 2503, 30: 901, 40: 601}[H]
    g = general_inputs(H, n, I=I, first=880000 + 10 * H + G)
    ou0, _, oit = oracle.solve_general(I, H, *[g[k] for k in GNAMES], nthreads=8)
    with _solver(H, G) as s:
        u0, it = _run(torch_cuda, s, g, I)
    assert np.array_equal(it, oit)
    assert np.abs(u0 - ou0).max() <= ATOL
    assert np.array_equal((ou0 == g["lo"]) | (ou0 == g["hi"]), (u0 == g["lo"]) | (u0 == g["hi"]))


@pytest.mark.parametrize("I", [1, 2])
@pytest.mark.parametrize("H,G", [(10, 4), (20, 8), (30, 8), (40, 4)])
def test_groupg_state_in_and_out(torch_cuda, oracle, I, H, G):
    """controls_inout + v_inout: warm start from random controls (shifted as operator() shifts them, mpc.h:231-232) and
    a random v; u0, the whole solved sequence, dlib's v and the iteration counts against the oracle."""
    from trajectory_controller_amd.synth import general_inputs
    n = 330 if H <= 20 else 140
    g = general_inputs(H, n, I=I, first=2468 + H)
    rng = np.random.default_rng(300 + H + I)
    cin = rng.uniform(-0.3, 0.3, size=(n, H, I))
    vin = rng.uniform(-0.3, 0.3, size=(n, H, I))
    g["lo"][::2] = -rng.uniform(0.02, 0.6, size=g["lo"][::2].shape)   # half of the boxes random: many warm starts lie OUTSIDE
    g["hi"][::2] = rng.uniform(0.02, 0.6, size=g["hi"][::2].shape)    # theirs (the stop test must then be dlib's own mask)
    ou0, cout, oit, vout = oracle.solve_general(I, H, *[g[k] for k in GNAMES], controls_in=cin, v_in=vin, want_v=True, nthreads=8)
    controls, vstate = _soa(cin), _soa(vin)
    with _solver(H, G) as s:
        s.set_profiling(True)
        u0, it = s.solve_batch_general(*[_soa(g[k]) for k in GNAMES], controls=controls, v_state=vstate, inputs=I, want_iters=True)
        assert s.last_kernel_times()[2] == GROUP
    assert np.array_equal(it, oit)
    assert np.abs(u0.T - ou0).max() <= ATOL
    assert np.abs(controls.T.reshape(n, H, I) - cout).max() <= ATOL
    assert np.abs(vstate.T.reshape(n, H, I) - vout).max() <= ATOL
    # controls only (v starts at zero and is not returned)
    ou0, cout, oit = oracle.solve_general(I, H, *[g[k] for k in GNAMES], controls_in=cin, nthreads=8)
    controls = _soa(cin)
    with _solver(H, G) as s:
        u0, it = s.solve_batch_general(*[_soa(g[k]) for k in GNAMES], controls=controls, inputs=I, want_iters=True)
    assert np.array_equal(it, oit) and np.abs(u0.T - ou0).max() <= ATOL
    assert np.abs(controls.T.reshape(n, H, I) - cout).max() <= ATOL


@pytest.mark.parametrize("I", [1, 2])
@pytest.mark.parametrize("H", [10, 20, 30, 40])
def test_groupg_rollout8_golden(torch_cuda, I, H):
    """8 controllers x 5 warm-started closed-loop steps (tpc_mpc_rollout keeps controls and v on the device between
    steps), expected values from real dlib: dlib's own acceptance threshold for such a loop is 1e-7 (test/mpc.cpp:312)."""
    g = load_golden(f"rollout8_I{I}_H{H}.npz")
    steps, n = int(g["steps"]), g["A"].shape[0]
    with _solver(H) as s:
        s.set_profiling(True)
        c, st, _ = s.rollout(steps, *[_soa(g[k]) for k in GNAMES], new_last_targets=_soa(g["new_last_targets"]), inputs=I)
        assert s.last_kernel_times()[2] == GROUP
    assert np.abs(c.T.reshape(n, steps, I) - g["controls"]).max() <= 1e-7


def test_groupg_rollout_kat(torch_cuda):
    """The reference's own known-answer scenario (dlib_files/dlib/test/mpc.cpp:266-317): mpc<2,1,30>, eps 1e-8."""
    g = load_golden("rollout_kat.npz")
    n = 3
    rep = lambda a: np.ascontiguousarray(np.repeat(np.asarray(a, dtype=np.float64).reshape(-1, 1), n, axis=1))
    with _solver(30, eps=1e-8, max_iter=10000) as s:
        c, st, it = s.rollout(30, rep(g["A"]), rep(g["B"]), rep(g["C"]), rep(g["Q"]), rep(g["R"]), rep(g["lo"]), rep(g["hi"]),
                              rep(g["x0"]), rep(g["targets0"].reshape(-1)), inputs=1, want_iters=True)
    for k in range(n):
        assert np.abs(c[:, k] - g["controls"][:, 0]).max() <= 1e-7


@pytest.mark.parametrize("I,H", [(2, 20), (1, 10), (2, 40)])
def test_groupg_hostile_batch_takes_the_exact_kernels(torch_cuda, oracle, I, H):
    """Pinned and one-sided boxes, a dead input column, Q == 0, large targets: the screen refuses the batch and the
    one-lane family's exact kernels solve it on the same records (iteration counts equal dlib's)."""
    from trajectory_controller_amd.synth import general_inputs
    from test_ub_model import hostile_general
    g = hostile_general(general_inputs(H, 440 if H <= 20 else 150, I=I, first=8100), I)
    ou0, _, oit = oracle.solve_general(I, H, *[g[k] for k in GNAMES], nthreads=8)
    with _solver(H) as s:
        u0, it = _run(torch_cuda, s, g, I)
        assert s.last_flags & ~2 == 0        # (long horizons: some instances end on the iteration cap, in dlib too)
    assert np.array_equal(it, oit) and np.abs(u0 - ou0).max() <= ATOL


def test_groupg_flags_phase_boundaries(torch_cuda, oracle):
    from trajectory_controller_amd.synth import general_inputs
    torch = torch_cuda
    I, H, n = 2, 10, 1200
    g = general_inputs(H, n, I=I, first=123)
    for smo, cap in ((0, 10000), (3, 10000), (50, 50), (50, 57), (7, 5)):
        ou0, _, oit = oracle.solve_general(I, H, *[g[k] for k in GNAMES], smo_iters=smo, max_iter=cap, nthreads=8)
        with _solver(H, 4, smo_iters=smo, max_iter=cap) as s:
            u0, it = _run(torch, s, g, I)
            assert bool(s.last_flags & 2) == bool((oit == cap).any())
        assert np.array_equal(it, oit), (smo, cap)
        assert np.abs(u0 - ou0).max() <= (ATOL if cap >= 10000 else 1e-6), (smo, cap)
    bad = {k: a.copy() for k, a in g.items()}
    bad["R"][5, 0] = 0.0          # min(R) > 0 violated
    bad["hi"][9, 1] = -1.0        # upper < lower
    with _solver(H, 4) as s:
        u0, it = _run(torch, s, bad, I)
        assert s.last_flags & 0x4
    assert np.all(u0[[5, 9]] == 0) and np.all(it[[5, 9]] == 0)
    keep = np.ones(n, bool); keep[[5, 9]] = False
    ou0, _, oit = oracle.solve_general(I, H, *[g[k] for k in GNAMES], nthreads=8)
    assert np.array_equal(it[keep], oit[keep]) and np.abs(u0[keep] - ou0[keep]).max() <= ATOL


@pytest.mark.parametrize("I,H,G,n", [(2, 10, 4, 3000), (2, 20, 4, 2000)])
def test_groupg_fp32(torch_cuda, oracle32, I, H, G, n):
    """fp32 (unpinned): against the float-typed restatement as a tolerance statement."""
    from trajectory_controller_amd.synth import general_inputs
    torch = torch_cuda
    g = {k: a.astype(np.float32) for k, a in general_inputs(H, n, I=I, first=300000).items()}
    ou0, _, oit = oracle32.solve_general(I, H, *[g[k] for k in GNAMES], nthreads=8)
    with _solver(H, G, dtype="f32") as s:
        u0, it = _run(torch, s, g, I, dtype=torch.float32)
    err = np.abs(u0 - ou0).max(axis=1)
    print(f"fp32 general GROUP I={I} H={H}: equal iteration counts {np.mean(it == oit):.4f}, median |du| {np.median(err):.2e}")
    assert np.isfinite(u0).all() and np.mean(it == oit) >= 0.4 and np.median(err) <= 1e-4
