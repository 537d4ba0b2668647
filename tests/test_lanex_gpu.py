"""The bit-exact LANE family's projected-gradient phase G lanes per instance (csrc/mpc_lanex.h, lanex_pg_kernel): compact
form, fp64, N = 10 / 20 / 30 / 40 (at N = 30 six lanes of a group of eight hold a chunk).  dlib's two recurrences (dlib_files/dlib/control/mpc.h:275-281) stay sequential -- handed from
chunk to chunk, lane to lane -- and everything off them is shared out over the lanes of a group: the SAME IEEE operations on
the same operands as one lane doing all N steps, so the bar is the LANE family's: bit-exact against real dlib and the
oracle (signed zeros distinguished), identical iteration counts.  Two users: AUTO's re-solve of instances a tolerance family
left on the iteration cap (tests/test_capped_gpu.py) and, below a measured batch size, the LANE family itself -- both paths
of that family (one lane per instance: `tpc_mpc_x_set_lanex_below(h, 0)`; G lanes: a huge limit) are held to dlib here.
Tolerance: none (max |du| == 0)."""
import numpy as np
import pytest

from conftest import bits_equal, load_golden

pytestmark = pytest.mark.gpu

ALWAYS, NEVER = 1 << 40, 0


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch


def _solve(torch, H, v, dy, dphi, below, **kw):
    from trajectory_controller_amd import MpcSolver
    tv, ty, tp = (torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to("cuda:0") for a in (v, dy, dphi))
    with MpcSolver(horizon=H, device=0, dtype="f64", algo="lane", **kw) as s:
        s._check(s._lib.tpc_mpc_x_set_lanex_below(s._h, below))
        f, r, it = s.solve_batch_compact(tv, ty, tp, want_iters=True)
        torch.cuda.synchronize()
        return f.cpu().numpy(), r.cpu().numpy(), it.cpu().numpy(), s.last_flags


@pytest.mark.parametrize("H", [10, 20, 40])
@pytest.mark.parametrize("below", [ALWAYS, NEVER], ids=["g_lanes", "one_lane"])
def test_lanex_golden(torch_cuda, H, below):
    """Real-dlib fixtures, both layouts of the bit-exact family."""
    g = load_golden(f"compact_H{H}.npz")
    f, r, it, _ = _solve(torch_cuda, H, g["v"], g["dy"], g["dphi"], below)
    assert bits_equal(f, g["front"]) and bits_equal(r, g["rear"])
    known = g["iters_lb"] >= 0
    assert np.all(it[known] >= g["iters_lb"][known])


@pytest.mark.parametrize("H,n", [(10, 3001), (20, 2000), (30, 1100), (40, 777), (20, 1), (30, 13), (40, 9), (10, 70)])
def test_lanex_vs_oracle(torch_cuda, oracle, H, n):
    """Seeded inputs, ragged batch sizes (partial groups and wavefronts): bits and iteration counts against the oracle."""
    from trajectory_controller_amd.synth import compact_inputs
    v, dy, dphi = compact_inputs(H, n, first=410000)
    of, orr, oit = oracle.solve_compact(H, v, dy, dphi, nthreads=8)
    f, r, it, _ = _solve(torch_cuda, H, v, dy, dphi, ALWAYS)
    assert np.array_equal(it, oit)
    assert bits_equal(f, of) and bits_equal(r, orr)


@pytest.mark.parametrize("H", [10, 20, 30, 40])
@pytest.mark.parametrize("kw", [dict(eps=1e-4, max_iter=300), dict(eps=0.05, smo_iters=0), dict(max_iter=60, smo_iters=50),
                                dict(lower=(-0.3, -0.2), upper=(0.25, 0.4)), dict(lower=(0.05, -0.3), upper=(0.3, -0.1)),
                                dict(weight_y=3.0, weight_phi=200.0, weight_steering_front=1e-6, weight_steering_rear=0.5)],
                         ids=["eps_cap", "no_cd", "cap_in_pg", "other_bounds", "start_outside_box", "weights"])
def test_lanex_knobs(torch_cuda, oracle, H, kw):
    """Knobs, iteration caps inside the phase, unequal bounds, a start point outside the box, other weights."""
    from trajectory_controller_amd.synth import compact_inputs
    v, dy, dphi = compact_inputs(H, 600, first=420000)
    okw = {}
    for name in ("eps", "max_iter", "smo_iters"):
        if name in kw:
            okw[name] = kw[name]
    if "lower" in kw:
        okw["lo"], okw["hi"] = kw["lower"], kw["upper"]
    if "weight_y" in kw:
        okw["weights"] = (kw["weight_y"], kw["weight_phi"], kw["weight_steering_front"], kw["weight_steering_rear"])
    of, orr, oit = oracle.solve_compact(H, v, dy, dphi, nthreads=8, **okw)
    f, r, it, _ = _solve(torch_cuda, H, v, dy, dphi, ALWAYS, **kw)
    assert np.array_equal(it, oit)
    assert bits_equal(f, of) and bits_equal(r, orr)


def test_lanex_edge_rows(torch_cuda):
    """NaN speed / offsets, zero target (every quantity a signed zero), saturating targets, v = 1e-3 and 50."""
    e = load_golden("compact_edge.npz")
    f, r, it, flags = _solve(torch_cuda, 20, e["v"], e["dy"], e["dphi"], ALWAYS)
    assert bits_equal(f, e["front_H20"]) and bits_equal(r, e["rear_H20"])
    assert flags & 1
    f1, r1, it1, _ = _solve(torch_cuda, 20, e["v"], e["dy"], e["dphi"], NEVER)
    assert bits_equal(f, f1) and bits_equal(r, r1) and np.array_equal(it, it1)


@pytest.mark.parametrize("H,n", [(40, 20000), (30, 12000), (20, 70000)])
def test_lanex_equals_one_lane_per_instance(torch_cuda, H, n):
    """More instances than the persistent grid holds groups (refill passes, the longest-first queue): the two layouts
    of the family agree in every bit and every iteration count."""
    from trajectory_controller_amd.synth import compact_inputs
    v, dy, dphi = compact_inputs(H, n, first=430000)
    a = _solve(torch_cuda, H, v, dy, dphi, ALWAYS)
    b = _solve(torch_cuda, H, v, dy, dphi, NEVER)
    assert bits_equal(a[0], b[0]) and bits_equal(a[1], b[1]) and np.array_equal(a[2], b[2]) and a[3] == b[3]


# ---------------------------------------------------------------------------------------------
# general form (lanexg_pg_kernel): per-instance A, B, C, Q, R, bounds, x0, per-step targets; one or two inputs; cold start

GNAMES = ["A", "B", "C", "Q", "R", "lo", "hi", "x0", "targets"]


def _solve_general(torch, I, H, g, below, **kw):
    from trajectory_controller_amd import MpcSolver
    n = g["A"].shape[0]
    dev = [torch.from_numpy(np.ascontiguousarray(np.asarray(g[k], dtype=np.float64).reshape(n, -1).T)).to("cuda:0") for k in GNAMES]
    with MpcSolver(horizon=H, device=0, dtype="f64", algo="lane", **kw) as s:
        s._check(s._lib.tpc_mpc_x_set_lanex_below(s._h, below))
        u0, it = s.solve_batch_general(*dev, inputs=I, want_iters=True)
        torch.cuda.synchronize()
        return u0.cpu().numpy().T, it.cpu().numpy(), s.last_flags


@pytest.mark.parametrize("I", [1, 2])
@pytest.mark.parametrize("H", [10, 20, 30, 40])
@pytest.mark.parametrize("below", [ALWAYS, NEVER], ids=["g_lanes", "one_lane"])
def test_lanex_general_golden(torch_cuda, I, H, below):
    g = load_golden(f"general_I{I}_H{H}.npz")
    u0, _, _ = _solve_general(torch_cuda, I, H, g, below)
    assert bits_equal(u0, g["u0"])


@pytest.mark.parametrize("I", [1, 2])
@pytest.mark.parametrize("H,n", [(10, 1501), (20, 900), (30, 500), (40, 333), (40, 3)])
def test_lanex_general_vs_oracle(torch_cuda, oracle, I, H, n):
    from trajectory_controller_amd.synth import general_inputs
    g = general_inputs(H, n, I=I, first=9100)
    ou0, _, oit = oracle.solve_general(I, H, *[g[k] for k in GNAMES], nthreads=8)
    u0, it, _ = _solve_general(torch_cuda, I, H, g, ALWAYS)
    assert np.array_equal(it, oit)
    assert bits_equal(u0, ou0)


@pytest.mark.parametrize("I", [1, 2])
@pytest.mark.parametrize("H", [20, 30, 40])
def test_lanex_general_hostile(torch_cuda, oracle, I, H):
    """Pinned and one-sided boxes, a dead input column (Q_diag == 0), Q = 0, large targets, a non-trivial A with negative
    and zero entries: bits and iteration counts against the oracle, and equal to the one-lane-per-instance kernels."""
    from trajectory_controller_amd.synth import general_inputs
    from test_ub_model import hostile_general
    g = hostile_general(general_inputs(H, 300, I=I, first=9300), I)
    g["A"] = g["A"].copy()
    g["A"][::7, 2] = -0.05      # a10 != 0
    g["A"][::11, 0] = -0.9      # a negative diagonal entry
    g["A"][::13, 3] = 0.0       # a zero one
    ou0, _, oit = oracle.solve_general(I, H, *[g[k] for k in GNAMES], nthreads=8, max_iter=1500)
    u0, it, fl = _solve_general(torch_cuda, I, H, g, ALWAYS, max_iter=1500)
    u1, it1, fl1 = _solve_general(torch_cuda, I, H, g, NEVER, max_iter=1500)
    assert np.array_equal(it, oit) and bits_equal(u0, ou0)
    assert np.array_equal(it, it1) and bits_equal(u0, u1) and fl == fl1


@pytest.mark.parametrize("I", [1, 2])
@pytest.mark.parametrize("H", [10, 20, 30, 40])
@pytest.mark.parametrize("below", [ALWAYS, NEVER], ids=["g_lanes", "one_lane"])
def test_lanex_general_state_in_out(torch_cuda, oracle, I, H, below):
    """controls_inout + v_inout (warm start from random controls -- some outside the box -- and a random v): outputs, the
    whole solved sequence, dlib's v and the iteration counts against the oracle, bit for bit, in both layouts of the
    family (G lanes per instance: lanexg_pg_kernel<STATE>; one lane per instance: lane_pg_kernel)."""
    from trajectory_controller_amd import MpcSolver
    from trajectory_controller_amd.synth import general_inputs
    n = 300 if H <= 20 else 121
    g = general_inputs(H, n, I=I, first=5200 + H)
    rng = np.random.default_rng(300 + H + I)
    cin = rng.uniform(-0.45, 0.45, size=(n, H, I))
    vin = rng.uniform(-0.3, 0.3, size=(n, H, I))
    ou0, ocout, oit, ovout = oracle.solve_general(I, H, *[g[k] for k in GNAMES], controls_in=cin, v_in=vin, want_v=True, nthreads=8)
    soa = lambda a: np.ascontiguousarray(np.asarray(a, dtype=np.float64).reshape(n, -1).T)
    controls, vstate = soa(cin), soa(vin)
    with MpcSolver(horizon=H, device=0, dtype="f64", algo="lane") as s:
        s._check(s._lib.tpc_mpc_x_set_lanex_below(s._h, below))
        u0, it = s.solve_batch_general(*[soa(g[k]) for k in GNAMES], controls=controls, v_state=vstate, inputs=I, want_iters=True)
    assert np.array_equal(it, oit)
    assert bits_equal(u0.T, ou0)
    assert bits_equal(controls.T.reshape(n, H, I), ocout)
    assert bits_equal(vstate.T.reshape(n, H, I), ovout)
