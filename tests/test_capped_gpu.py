"""`-m gpu` tests of AUTO's parity guarantee for instances that end on the iteration cap (include/tpc_mpc.h, AUTO).

An instance that runs into max_iter has not converged; over thousands of iterations of an ill-conditioned problem
the tolerance families' rounding differences grow (profiles/r03_fuzz_lane_fma.txt: 2.3e-5 at N = 40 under adversarial
parameters, profiles/r04_fuzz_capped.txt: 7e-8 on the set used below).  Under AUTO such instances are solved once more
in dlib's own operation order (lane_cd_kernel, RESOLVE), so what the call returns for them is dlib's BITS whichever
family solved the batch -- WAVE, GROUP or LANE_FMA, compact or general form, batch or solve_one -- and every instance
is within the 1e-6 target.  TPC_MPC_PARAM_FAST_CAPPED keeps the tolerance family's answer.
"""
import numpy as np
import pytest

from conftest import bits_equal

pytestmark = pytest.mark.gpu

WAVE, LANE, LANE_FMA, GROUP = 1, 2, 3, 4
# the worst set of profiles/r04_fuzz_capped.txt: N = 40, 544 of 750 instances on the cap
HOSTILE = dict(weight_y=17.66698177195007, weight_phi=72.39497127615921, weight_steering_front=0.00021630579900267817,
               weight_steering_rear=0.045780005743250786, lower=(-0.40132006097329725, -0.13466087933638904),
               upper=(0.13466087933638904, 0.40132006097329725), step_size=0.22085310545663503,
               wheelbase=0.12366626463869, eps=0.004058013395383058, max_iter=10000, smo_iters=7)


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch


def _oracle_kw(kw):
    return dict(weights=(kw["weight_y"], kw["weight_phi"], kw["weight_steering_front"], kw["weight_steering_rear"]),
                T=kw["step_size"], l=kw["wheelbase"], lo=kw["lower"], hi=kw["upper"], eps=kw["eps"],
                max_iter=kw["max_iter"], smo_iters=kw["smo_iters"])


def _solve(torch, H, v, dy, dphi, algo="auto", expect=None, **kw):
    from trajectory_controller_amd import MpcSolver
    tv, ty, tp = (torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0") for a in (v, dy, dphi))
    with MpcSolver(horizon=H, device=0, algo=algo, **kw) as s:
        s.set_profiling(True)
        f, r, it = s.solve_batch_compact(tv, ty, tp, want_iters=True)
        torch.cuda.synchronize()
        if expect is not None:
            assert s.last_kernel_times()[2] == expect
        flags = s.last_flags
    return f.cpu().numpy(), r.cpu().numpy(), it.cpu().numpy(), flags


def test_hostile_set_capped_instances_come_back_with_dlibs_bits(torch_cuda, oracle):
    from trajectory_controller_amd import FLAG_MAX_ITER, capi
    from trajectory_controller_amd.synth import compact_inputs
    H, n = 40, 750
    v, dy, dphi = compact_inputs(H, n, first=834746896)
    of, orr, oit = oracle.solve_compact(H, v, dy, dphi, nthreads=8, **_oracle_kw(HOSTILE))
    capped = oit == HOSTILE["max_iter"]
    assert capped.sum() > 400
    den = np.maximum(np.maximum(np.abs(of), np.abs(orr)), 0.13)
    f, r, it, flags = _solve(torch_cuda, H, v, dy, dphi, expect=WAVE, **HOSTILE)
    assert flags & FLAG_MAX_ITER and np.array_equal(it, oit)
    assert bits_equal(f[capped], of[capped]) and bits_equal(r[capped], orr[capped])
    assert (np.maximum(np.abs(f - of), np.abs(r - orr)) / den).max() <= 1e-6
    # switched off: the tolerance family's own answer (close, but not dlib's bits on the capped instances)
    f2, r2, it2, _ = _solve(torch_cuda, H, v, dy, dphi, expect=WAVE, options=capi.PARAM_FAST_CAPPED, **HOSTILE)
    assert np.array_equal(it2, oit)
    assert not (bits_equal(f2[capped], of[capped]) and bits_equal(r2[capped], orr[capped]))
    assert np.array_equal(f2[~capped], f[~capped]) and np.array_equal(r2[~capped], r[~capped])
    # an explicitly demanded tolerance family is taken at its word
    f3, r3, _, _ = _solve(torch_cuda, H, v, dy, dphi, algo="wave", **HOSTILE)
    assert np.array_equal(f3, f2) and np.array_equal(r3, r2)


@pytest.mark.parametrize("H,n,cap,family", [(40, 8192, 10000, GROUP), (20, 16384, 1500, GROUP), (10, 100000, 300, LANE_FMA),
                                            (20, 3000, 1000, WAVE), (5, 40000, 60, LANE_FMA), (30, 1024, 3000, WAVE)])
def test_capped_instances_are_bit_exact_in_every_family(torch_cuda, oracle, H, n, cap, family):
    """BASELINE inputs with a cap low enough that a good part of the batch runs into it, at batch sizes that send AUTO
    to each of its tolerance families: the capped instances equal the oracle bit for bit, the rest is within 1e-9,
    iteration counts are the oracle's everywhere."""
    from trajectory_controller_amd import FLAG_MAX_ITER
    from trajectory_controller_amd.synth import compact_inputs
    v, dy, dphi = compact_inputs(H, n, first=123457)
    of, orr, oit = oracle.solve_compact(H, v, dy, dphi, max_iter=cap, nthreads=8)
    capped = oit == cap
    assert 0 < capped.sum() < n
    f, r, it, flags = _solve(torch_cuda, H, v, dy, dphi, expect=family, max_iter=cap)
    assert flags & FLAG_MAX_ITER and np.array_equal(it, oit)
    assert bits_equal(f[capped], of[capped]) and bits_equal(r[capped], orr[capped])
    assert max(np.abs(f - of).max(), np.abs(r - orr).max()) <= 1e-9


def test_no_capped_instance_no_change(torch_cuda, oracle):
    """The usual case: nothing ends on the cap, the re-solve kernel leaves at once and the call returns what the
    tolerance family computed."""
    from trajectory_controller_amd import capi
    from trajectory_controller_amd.synth import compact_inputs
    H, n = 20, 16384
    v, dy, dphi = compact_inputs(H, n)
    f, r, it, flags = _solve(torch_cuda, H, v, dy, dphi, expect=GROUP)
    f2, r2, it2, _ = _solve(torch_cuda, H, v, dy, dphi, expect=GROUP, options=capi.PARAM_FAST_CAPPED)
    assert flags == 0 and np.array_equal(f, f2) and np.array_equal(r, r2) and np.array_equal(it, it2)


@pytest.mark.parametrize("I,n,family", [(1, 2000, WAVE), (2, 2000, WAVE), (2, 16384, GROUP), (1, 300000, LANE_FMA)])
def test_general_form_capped_instances(torch_cuda, oracle, I, n, family):
    """The general form (per-instance model, per-step targets, cold start) through AUTO, at batch sizes that send it to
    each of its tolerance families."""
    import torch
    from trajectory_controller_amd import MpcSolver
    from trajectory_controller_amd.synth import general_inputs
    H, cap = 20, 400
    names = ["A", "B", "C", "Q", "R", "lo", "hi", "x0", "targets"]
    g = general_inputs(H, n, I=I, first=4711)
    ou0, _, oit = oracle.solve_general(I, H, *[g[k] for k in names], max_iter=cap, nthreads=8)
    capped = oit == cap
    assert 0 < capped.sum() < n
    dev = [torch.from_numpy(np.ascontiguousarray(g[k].reshape(n, -1).T)).to("cuda:0") for k in names]
    with MpcSolver(horizon=H, device=0, max_iter=cap) as s:
        s.set_profiling(True)
        u0, it = s.solve_batch_general(*dev, inputs=I, want_iters=True)
        torch.cuda.synchronize()
        assert s.last_kernel_times()[2] == family
    u0, it = u0.cpu().numpy().T, it.cpu().numpy()
    assert np.array_equal(it, oit)
    assert bits_equal(u0[capped], ou0[capped]) and np.abs(u0 - ou0).max() <= 1e-9


def test_solve_one_capped(torch_cuda, oracle):
    """The call that replaces mpcControllerTobi: a solve the resident wavefront leaves on the cap is solved once more
    bit-exactly (one LANE launch), the flag stays raised."""
    from trajectory_controller_amd import FLAG_MAX_ITER, MpcSolver
    of, orr, oit = oracle.solve_compact(20, [2.0, 50.0], [-0.2, 0.3], [0.1, 0.2], max_iter=200)
    with MpcSolver(horizon=20, device=0, max_iter=200) as s:
        for k in range(2):
            f, r = s.solve_one([2.0, 50.0][k], [-0.2, 0.3][k], [0.1, 0.2][k])
            assert s.last_solve_one_flags() == (FLAG_MAX_ITER, 200) and oit[k] == 200
            assert bits_equal([f, r], [of[k], orr[k]])


def _lambda(H, v, T=0.1, l=0.21, q=(20.0, 7.0), r=(0.0005, 10.0)):
    """dlib's trace bound (mpc.h:116-123) for the compact model, on the host: what AUTO's presolve predicts from"""
    out = np.empty(len(v))
    for i, x in enumerate(v):
        a, c = T * x, T * x / l
        A = np.array([[1.0, a], [0.0, 1.0]])
        B = np.array([[0.0, a], [c, -c]])
        lam, Tm = (r[0] + r[1]) * H, np.diag(q).astype(float)
        for _ in range(H):
            P = B.T @ Tm @ B
            lam += P[0, 0] + P[1, 1]
            Tm = A.T @ Tm @ A + np.diag(q)
        out[i] = lam
    return out


@pytest.mark.parametrize("eps, expect_missed, expect_spare", [(0.01, False, True), (1e-5, True, False), (0.5, False, True)])
def test_presolve_prediction_wrong_in_both_directions(torch_cuda, oracle, eps, expect_missed, expect_spare):
    """AUTO's presolve (tpc_mpc_api.cpp: presolve_begin) takes the instances with lambda >= (max_iter / 7)^2 to the
    bit-exact kernels BEFORE the tolerance family runs, on a side stream.  That is a prediction: with dlib's eps it names
    every capped instance of the N = 40 stream and some that converge after all (spare); with a tiny eps far more instances
    cap than it names (missed: the flag-driven second pass must catch them); with a huge eps nothing caps and everything it
    took is spare.  In every case: each capped instance and each instance the prediction took comes back with dlib's
    BITS, every other one within 1e-9, every iteration count equal -- the same outcome as before the presolve existed."""
    from trajectory_controller_amd import FLAG_MAX_ITER
    from trajectory_controller_amd.synth import compact_inputs
    H, n, max_iter = 40, 4096, 10000
    v, dy, dphi = compact_inputs(H, n, first=4242)
    of, orr, oit = oracle.solve_compact(H, v, dy, dphi, nthreads=8, eps=eps, max_iter=max_iter)
    capped = oit == max_iter
    taken = _lambda(H, v) >= (max_iter / 7.0) ** 2
    assert taken.sum() > 100
    assert bool((capped & ~taken).any()) == expect_missed and bool((taken & ~capped).any()) == expect_spare
    f, r, it, flags = _solve(torch_cuda, H, v, dy, dphi, expect=GROUP, eps=eps, max_iter=max_iter)
    assert np.array_equal(it, oit)
    exact = capped | (taken & (oit > 50))   # (what stops inside the coordinate-descent phase is not queued by either pass)
    assert bits_equal(f[exact], of[exact]) and bits_equal(r[exact], orr[exact])
    assert max(np.abs(f - of).max(), np.abs(r - orr).max()) <= 1e-9
    assert bool(flags & FLAG_MAX_ITER) == bool(capped.any())


def test_reserve_covers_the_presolve(torch_cuda):
    """tpc_mpc_reserve sizes everything a later solve of that shape needs -- AUTO's presolve scratch, side stream and
    events included (tpc_mpc_api.cpp) -- so that no solve allocates (an allocation synchronises the device): free device
    memory does not move across the first N = 40 AUTO solve after a reserve, and it did move in the reserve."""
    from trajectory_controller_amd import MpcSolver
    from trajectory_controller_amd.synth import compact_inputs
    torch = torch_cuda
    H, n = 40, 16384
    v, dy, dphi = (torch.from_numpy(a).cuda() for a in compact_inputs(H, n, first=17))
    with MpcSolver(horizon=H, device=0, algo="auto") as s:
        s.set_profiling(True)
        torch.cuda.synchronize()
        free0 = torch.cuda.mem_get_info()[0]
        s.reserve(n)
        torch.cuda.synchronize()
        free1 = torch.cuda.mem_get_info()[0]
        assert free0 - free1 >= 2 * n * 8 * 2 * H   # at least the records of the main scratch and of the presolve's own
        f, r, it = s.solve_batch_compact(v, dy, dphi, want_iters=True)
        torch.cuda.synchronize()
        free2 = torch.cuda.mem_get_info()[0]
        grown_by_torch = f.numel() * 8 * 2 + it.numel() * 4 + (4 << 20)   # the outputs torch allocated (its pool may grow in 2 MiB blocks)
        assert free1 - free2 <= grown_by_torch, (free1 - free2, grown_by_torch)
        assert s.last_kernel_times()[2] == GROUP
