"""`-m gpu` tests of the C-ABI surface added in round 2, all through libtpc_mpc.so:
shards with ld > n and a shifted base pointer, allocation failure as a status code, the mixed-horizon
entry, one handle on two streams, the sharded entry (world of one through RCCL-free code, two ranks of
the PRODUCT solver under torch.distributed/gloo), per-step targets from the trajectory, and the
resident solve_one kernel."""
import ctypes as C
import os
import subprocess
import sys
import time

import numpy as np
import pytest

from conftest import bits_equal, load_golden

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GEN_NAMES = ["A", "B", "C", "Q", "R", "lo", "hi", "x0", "targets"]
SENT = 777.25   # marks memory the library must not touch


def _solver(H, algo="lane", dtype="f64", **kw):
    from trajectory_controller_amd import MpcSolver
    return MpcSolver(horizon=H, device=0, dtype=dtype, algo=algo, **kw)


def _soa(a):
    a = np.asarray(a)
    return np.ascontiguousarray(a.reshape(a.shape[0], -1).T)


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch


# ---------------------------------------------------------------------------------------------
# shards: base + k0 with the full batch's ld (include/tpc_mpc.h, Conventions)

def _general_shard_call(s, mem, torch, I, H, n_full, k0, n, arrays, controls, vstate):
    """Calls tpc_mpc_solve_batch_general on columns [k0, k0+n) of full-width [comps, n_full] arrays."""
    from trajectory_controller_amd import capi
    es = 8
    keep = []

    def ptr(a):
        if a is None:
            return None
        if mem == capi.DEVICE:
            keep.append(a)
            return a.data_ptr() + k0 * es
        keep.append(a)
        return a.ctypes.data + k0 * es
    u0 = np.full((I, n_full), SENT)
    iters = np.full(n_full, -5, dtype=np.int32)
    if mem == capi.DEVICE:
        u0, iters = torch.from_numpy(u0).cuda(), torch.from_numpy(iters).cuda()
    io = capi.GeneralIO(inputs=I, n=n, ld=n_full, A=ptr(arrays[0]), B=ptr(arrays[1]), C=ptr(arrays[2]),
                        Q=ptr(arrays[3]), R=ptr(arrays[4]), lower=ptr(arrays[5]), upper=ptr(arrays[6]),
                        x0=ptr(arrays[7]), targets=ptr(arrays[8]), controls_inout=ptr(controls),
                        v_inout=ptr(vstate), u0=ptr(u0),
                        iters=(iters.data_ptr() if mem == capi.DEVICE else iters.ctypes.data) + 4 * k0)
    flags = C.c_uint32(0)
    s._check(s._lib.tpc_mpc_solve_batch_general(s._h, C.byref(s.params), C.byref(io), C.byref(flags), mem, None))
    if mem == capi.DEVICE:
        torch.cuda.synchronize()
        u0, iters = u0.cpu().numpy(), iters.cpu().numpy()
    return u0, iters


@pytest.mark.parametrize("mem", ["host", "device"])
@pytest.mark.parametrize("algo", ["lane", "wave"])
def test_general_shard_of_wider_batch(torch_cuda, oracle, mem, algo):
    """ld > n with a shifted base pointer: only the shard's n columns are read and written (the
    round-1 library copied whole comps*ld slabs: it read k0 elements past every array and wrote
    staging garbage over the neighbouring shards' columns)."""
    from trajectory_controller_amd import capi
    from trajectory_controller_amd.synth import general_inputs
    torch = torch_cuda
    I, H, n_full, k0, n = 2, 10, 700, 130, 333
    g = general_inputs(H, n_full, I=I, first=9)
    rng = np.random.default_rng(8)
    cin = rng.uniform(-0.3, 0.3, size=(n_full, H, I))
    vin = rng.uniform(-0.3, 0.3, size=(n_full, H, I))
    sl = slice(k0, k0 + n)
    u0, cout, it, vout = oracle.solve_general(I, H, *[g[k][sl] for k in GEN_NAMES], controls_in=cin[sl],
                                              v_in=vin[sl], want_v=True, nthreads=8)
    arrays = [_soa(g[k]) for k in GEN_NAMES]
    controls, vstate = _soa(cin), _soa(vin)
    c_before, v_before = controls.copy(), vstate.copy()
    m = capi.HOST if mem == "host" else capi.DEVICE
    if m == capi.DEVICE:
        arrays = [torch.from_numpy(a).cuda() for a in arrays]
        controls, vstate = torch.from_numpy(controls).cuda(), torch.from_numpy(vstate).cuda()
    with _solver(H, algo) as s:
        gu0, git = _general_shard_call(s, m, torch, I, H, n_full, k0, n, arrays, controls, vstate)
    if m == capi.DEVICE:
        controls, vstate = controls.cpu().numpy(), vstate.cpu().numpy()
    outside = np.ones(n_full, dtype=bool)
    outside[sl] = False
    assert np.all(gu0[:, outside] == SENT) and np.all(git[outside] == -5)          # neighbours untouched
    assert np.array_equal(controls[:, outside], c_before[:, outside])
    assert np.array_equal(vstate[:, outside], v_before[:, outside])
    assert np.array_equal(git[sl], it)
    got_c = controls[:, sl].T.reshape(n, H, I)
    got_v = vstate[:, sl].T.reshape(n, H, I)
    if algo == "lane":
        assert bits_equal(gu0[:, sl].T, u0) and bits_equal(got_c, cout) and bits_equal(got_v, vout)
    else:
        assert np.abs(gu0[:, sl].T - u0).max() <= 1e-9 and np.abs(got_c - cout).max() <= 1e-9


@pytest.mark.parametrize("mem", ["host", "device"])
def test_rollout_shard_of_wider_batch(torch_cuda, mem):
    """tpc_mpc_rollout on columns [k0, k0+n) of wider arrays: equals the rollout of the same
    instances passed contiguously, and nothing outside the shard's columns is written."""
    from trajectory_controller_amd import capi
    from trajectory_controller_amd.synth import general_inputs
    torch = torch_cuda
    I, H, n_full, k0, n, steps = 2, 5, 300, 70, 129, 4
    g = general_inputs(H, n_full, I=I, first=3)
    rng = np.random.default_rng(1)
    nlt = _soa(g["targets"][:, -1:, :] + rng.uniform(-0.05, 0.05, size=(n_full, steps, 2)))
    sl = slice(k0, k0 + n)
    with _solver(H, "lane") as s:
        want_c, want_s, want_i = s.rollout(steps, *[np.ascontiguousarray(_soa(g[k])[:, sl]) for k in GEN_NAMES],
                                           new_last_targets=np.ascontiguousarray(nlt[:, sl]), inputs=I,
                                           want_iters=True)
        arrays = [_soa(g[k]) for k in GEN_NAMES]
        c_out = np.full((steps * I, n_full), SENT)
        s_out = np.full((steps * 2, n_full), SENT)
        i_out = np.full((steps, n_full), -5, dtype=np.int32)
        controls = np.full((H * I, n_full), 0.0)
        m = capi.HOST if mem == "host" else capi.DEVICE
        bufs = arrays + [nlt, c_out, s_out, i_out, controls]
        if m == capi.DEVICE:
            bufs = [torch.from_numpy(a).cuda() for a in bufs]
        base = (lambda a, e=8: a.data_ptr() + k0 * e) if m == capi.DEVICE else (lambda a, e=8: a.ctypes.data + k0 * e)
        io = capi.GeneralIO(inputs=I, n=n, ld=n_full, A=base(bufs[0]), B=base(bufs[1]), C=base(bufs[2]),
                            Q=base(bufs[3]), R=base(bufs[4]), lower=base(bufs[5]), upper=base(bufs[6]),
                            x0=base(bufs[7]), targets=base(bufs[8]), controls_inout=base(bufs[13]), v_inout=None,
                            u0=None, iters=None)
        flags = C.c_uint32(0)
        s._check(s._lib.tpc_mpc_rollout(s._h, C.byref(s.params), C.byref(io), steps, base(bufs[9]), base(bufs[10]),
                                        base(bufs[11]), base(bufs[12], 4), C.byref(flags), m, None))
        if m == capi.DEVICE:
            torch.cuda.synchronize()
            bufs = [a.cpu().numpy() for a in bufs]
    c_out, s_out, i_out, controls = bufs[10], bufs[11], bufs[12], bufs[13]
    outside = np.ones(n_full, dtype=bool)
    outside[sl] = False
    assert np.all(c_out[:, outside] == SENT) and np.all(s_out[:, outside] == SENT) and np.all(i_out[:, outside] == -5)
    assert np.all(controls[:, outside] == 0.0)
    assert bits_equal(c_out[:, sl], want_c) and bits_equal(s_out[:, sl], want_s) and np.array_equal(i_out[:, sl], want_i)
    for k, a in zip(GEN_NAMES, bufs[:9]):           # inputs are inputs
        assert np.array_equal(a, _soa(g[k])), k


def test_allocation_failure_is_a_status_code(torch_cuda):
    """A reservation no GPU can satisfy (about 0.8 TB of scratch) comes back as TPC_MPC_ERR_ALLOC, and
    the handle keeps working."""
    from trajectory_controller_amd import capi
    with _solver(20, "lane") as s:
        rc = s._lib.tpc_mpc_reserve(s._h, C.byref(s.params), 2**31 - 1, capi.DEVICE)
        assert rc == 8, rc
        assert b"hipMalloc" in s._lib.tpc_mpc_last_error(s._h)
        f, r = s.solve_batch_compact(np.array([1.0] * 70), np.array([0.1] * 70), np.array([0.05] * 70))
        assert np.all(f == f[0]) and abs(f[0]) > 1e-3


def test_build_info_names_the_schedulers(torch_cuda):
    from trajectory_controller_amd import MpcSolver
    info = MpcSolver.build_info()
    print(info)
    assert "abi 5" in info and all(f"h{h}[sched=default]" in info for h in (4, 5, 10, 20, 30, 40))


# ---------------------------------------------------------------------------------------------
# mixed horizons through the C entry

def test_mixed_entry_host_and_bad_horizon(torch_cuda, oracle):
    from trajectory_controller_amd import TpcMpcError
    from trajectory_controller_amd.synth import compact_inputs
    Hs = np.array([4, 5, 10, 20, 30, 40])
    n = 3000
    rng = np.random.default_rng(2)
    hz = Hs[rng.integers(0, len(Hs), size=n)]
    hz[:5] = 40
    v, dy, dphi = compact_inputs(20, n, first=60000)
    with _solver(20, "lane") as s:
        f, r, it = s.solve_batch_compact_mixed(hz, v, dy, dphi, want_iters=True)      # host arrays
        for H in Hs:
            sel = hz == H
            of, orr, oit = oracle.solve_compact(int(H), v[sel], dy[sel], dphi[sel], nthreads=8)
            assert np.array_equal(it[sel], oit), H
            assert bits_equal(f[sel], of) and bits_equal(r[sel], orr), H
        bad = hz.copy()
        bad[1234] = 7
        with pytest.raises(TpcMpcError) as e:
            s.solve_batch_compact_mixed(bad, v, dy, dphi)
        assert e.value.status == 4
        f2, r2 = s.solve_batch_compact_mixed(hz, v, dy, dphi)                          # handle still fine
        assert bits_equal(f2, f) and bits_equal(r2, r)


def test_one_handle_two_streams(torch_cuda):
    """Back-to-back solves of ONE handle on two non-blocking streams: the library orders them (an
    event between the streams), so both equal the single-stream results."""
    from trajectory_controller_amd.synth import compact_inputs
    torch = torch_cuda
    H, n = 20, 70000
    sets = [[torch.from_numpy(a).cuda() for a in compact_inputs(H, n, first=1000 * b)] for b in range(2)]
    with _solver(H, "lane") as s:
        want = [s.solve_batch_compact(*x, want_iters=True) for x in sets]
        torch.cuda.synchronize()
        streams = [torch.cuda.Stream(), torch.cuda.Stream()]
        got = []
        for k in range(6):
            with torch.cuda.stream(streams[k % 2]):
                got.append(s.solve_batch_compact(*sets[k % 2], want_iters=True, want_flags=False))
        torch.cuda.synchronize()
    for k in range(6):
        for a, b in zip(got[k], want[k % 2]):
            assert torch.equal(a, b)


# ---------------------------------------------------------------------------------------------
# sharding

def test_sharded_entry_world_of_one(torch_cuda):
    """tpc_mpc_solve_batch_compact_sharded on a handle without a communicator (a world of one) and
    after tpc_mpc_comm_init_rank(world = 1): the full batch, straight into the full-size outputs."""
    from trajectory_controller_amd import capi
    from trajectory_controller_amd.synth import compact_inputs
    torch = torch_cuda
    H, n = 10, 5001
    g = load_golden(f"compact_H{H}.npz")
    v, dy, dphi = (torch.from_numpy(a).cuda() for a in compact_inputs(H, n))
    first, count = C.c_int64(), C.c_int64()
    lib = capi.load_library()
    assert lib.tpc_mpc_shard_range(1001, 1, 2, C.byref(first), C.byref(count)) == 0
    assert (first.value, count.value) == (501, 500)
    with _solver(H, "lane") as s:
        f, r = s.solve_batch_compact_sharded(n, v, dy, dphi)
        s.comm_init(b"\0" * 128, 0, 1)
        f2, r2, it = s.solve_batch_compact_sharded(n, v, dy, dphi, want_iters=True, want_flags=True)
        torch.cuda.synchronize()
    assert bits_equal(f[:1024].cpu().numpy(), g["front"]) and bits_equal(r[:1024].cpu().numpy(), g["rear"])
    assert torch.equal(f, f2) and torch.equal(r, r2) and int(it.min()) >= 0


@pytest.mark.parametrize("ragged", [False, True])
def test_sharded_entry_through_rccl_world_of_one(torch_cuda, ragged):
    """The RCCL calls themselves on the one-GPU box: after tpc_mpc_comm_test_mode(force_communicator = 1)
    tpc_mpc_comm_init_rank builds a real one-rank communicator (dlopen, ncclGetUniqueId, ncclCommInitRank), and
    the sharded solve then runs its grouped in-place ncclAllGather or (force_ragged = 1) the per-owner
    ncclBroadcast of ragged batches on the caller's stream.  Results = the plain solve."""
    from trajectory_controller_amd import MpcSolver
    from trajectory_controller_amd.synth import compact_inputs
    torch = torch_cuda
    H, n = 10, 4097
    v, dy, dphi = (torch.from_numpy(a).cuda() for a in compact_inputs(H, n, first=11))
    with _solver(H, "lane") as s:
        want = s.solve_batch_compact(v, dy, dphi)
        s.comm_test_mode(True, ragged)
        s.comm_init(MpcSolver.comm_unique_id(), 0, 1)
        stream = torch.cuda.Stream()
        with torch.cuda.stream(stream):
            f, r = s.solve_batch_compact_sharded(n, v, dy, dphi)
        stream.synchronize()
    assert torch.equal(f, want[0]) and torch.equal(r, want[1])


def test_interleaved_split_through_rccl_world_of_one(torch_cuda):
    """tpc_mpc_solve_batch_compact_sharded_split / tpc_mpc_gather_shards_split with TPC_MPC_SPLIT_INTERLEAVED through a real
    one-rank RCCL communicator: the shard is solved into the handle's staging array, all-gathered there and written back
    in instance order by unpermute_kernel (a world of one: the identity, but every launch of the path runs); without a
    communicator the shard is solved straight into the outputs.  Results = the plain solve; an unknown split is refused."""
    from trajectory_controller_amd import MpcSolver, capi
    from trajectory_controller_amd.synth import compact_inputs
    torch = torch_cuda
    H, n = 10, 4097
    v, dy, dphi = (torch.from_numpy(a).cuda() for a in compact_inputs(H, n, first=23))
    with _solver(H, "lane") as s:
        want = s.solve_batch_compact(v, dy, dphi, want_iters=True)
        f0, r0 = s.solve_batch_compact_sharded(n, v, dy, dphi, split="interleaved")     # no communicator
        assert s.shard_map(n, "interleaved") == (0, n, 1)
        s.comm_test_mode(True, False)
        s.comm_init(MpcSolver.comm_unique_id(), 0, 1)
        stream = torch.cuda.Stream()
        with torch.cuda.stream(stream):
            f, r, it = s.solve_batch_compact_sharded(n, v, dy, dphi, split="interleaved", want_iters=True)
            rows = torch.stack([want[0], want[1]]).clone()
            s.gather_shards(n, rows, split="interleaved")
            it2 = want[2].clone()
            s.gather_shards(n, it2, split="interleaved")                                # 4-byte elements
        stream.synchronize()
        p = s._params()
        import ctypes as C
        with pytest.raises(capi.TpcMpcError):
            s._check(s._lib.tpc_mpc_solve_batch_compact_sharded_split(s._h, C.byref(p), n, 2, v.data_ptr(), dy.data_ptr(),
                                                                      dphi.data_ptr(), f.data_ptr(), r.data_ptr(), None, None, None))
    for a, b in ((f0, want[0]), (r0, want[1]), (f, want[0]), (r, want[1]), (it, want[2]), (rows[0], want[0]), (rows[1], want[1]),
                 (it2, want[2])):
        assert torch.equal(a, b)


@pytest.mark.parametrize("ragged", [False, True])
def test_general_sharded_entry(torch_cuda, oracle, ragged):
    """tpc_mpc_solve_batch_general_sharded: a world of one without a communicator, then through a real one-rank RCCL
    communicator (grouped in-place ncclAllGather per row of u0, or the per-owner ncclBroadcast form): the full batch,
    bit-identical to the plain general solve and to the oracle (LANE family)."""
    from trajectory_controller_amd import MpcSolver
    from trajectory_controller_amd.synth import general_inputs
    torch = torch_cuda
    I, H, n = 2, 10, 2049
    g = general_inputs(H, n, I=I, first=321)
    names = ["A", "B", "C", "Q", "R", "lo", "hi", "x0", "targets"]
    ou0, _, oit = oracle.solve_general(I, H, *[g[k] for k in names], nthreads=8)
    dev = [torch.from_numpy(np.ascontiguousarray(g[k].reshape(n, -1).T)).cuda() for k in names]
    with _solver(H, "lane") as s:
        want = s.solve_batch_general(*dev, inputs=I)
        u0 = s.solve_batch_general_sharded(*dev, inputs=I)
        s.comm_test_mode(True, ragged)
        s.comm_init(MpcSolver.comm_unique_id(), 0, 1)
        stream = torch.cuda.Stream()
        with torch.cuda.stream(stream):
            u1, it = s.solve_batch_general_sharded(*dev, inputs=I, want_iters=True)
        stream.synchronize()
        assert s.last_flags == 0
    assert torch.equal(u0, want) and torch.equal(u1, want)
    assert bits_equal(want.cpu().numpy().T, ou0) and np.array_equal(it.cpu().numpy(), oit)


@pytest.mark.parametrize("ragged", [False, True])
def test_gather_shards_behind_the_mixed_entry(torch_cuda, ragged):
    """tpc_mpc_gather_shards: the exchange by itself, for entries that have no sharded form of their own -- here a mixed-
    horizon batch solved into full-size outputs and exchanged through a real one-rank RCCL communicator (grouped in-place
    ncclAllGather per row, or the per-owner ncclBroadcast form), fp64 rows and an int32 row (moved as bit patterns).  A
    world of one must find its rows unchanged; without a communicator the call is a no-op; bad arguments are refused."""
    from trajectory_controller_amd import MpcSolver, capi
    from trajectory_controller_amd.synth import compact_inputs
    torch = torch_cuda
    n = 4099
    hz = np.resize(np.array([5, 10, 20], dtype=np.int32), n)
    v, dy, dphi = (torch.from_numpy(a).cuda() for a in compact_inputs(20, n, first=77))
    with _solver(20, "auto") as s:
        f, r, it = s.solve_batch_compact_mixed(hz, v, dy, dphi, want_iters=True)
        want = (f.clone(), r.clone(), it.clone())
        s.gather_shards(n, f, r)                       # no communicator: nothing to do
        assert s.shard_range(n) == (0, n)
        s.comm_test_mode(True, ragged)
        s.comm_init(MpcSolver.comm_unique_id(), 0, 1)
        fr = torch.stack([f, r])                       # a 2-D tensor: its rows are exchanged
        stream = torch.cuda.Stream()
        stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(stream):
            s.gather_shards(n, fr)
            s.gather_shards(n, f, r)
            s.gather_shards(n, it)                     # 4-byte elements
        stream.synchronize()
        assert torch.equal(fr[0], want[0]) and torch.equal(fr[1], want[1])
        assert torch.equal(f, want[0]) and torch.equal(r, want[1]) and torch.equal(it, want[2])
        with pytest.raises(capi.TpcMpcError):
            s._check(s._lib.tpc_mpc_gather_shards(s._h, n, None, 2, 8, None))
        with pytest.raises(capi.TpcMpcError):
            import ctypes as C
            table = (C.c_void_p * 1)(f.data_ptr())
            s._check(s._lib.tpc_mpc_gather_shards(s._h, n, table, 1, 2, None))


_RANK_WORKER = r'''
import os, sys
sys.path.insert(0, {root!r})
import numpy as np, torch, torch.distributed as dist
from trajectory_controller_amd import MpcSolver
from trajectory_controller_amd.shard import solve_sharded, shard_range
from trajectory_controller_amd.synth import compact_inputs
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
H, n = 20, 2049                                   # ragged: 1025 + 1024; the first 1024 are the golden ones
dev = torch.device("cuda:0")                      # both ranks share the one card of the test box
v, dy, dphi = (torch.from_numpy(a).to(dev) for a in compact_inputs(H, n))
with MpcSolver(horizon=H, device=0, algo="lane") as s:
    def solve(a, b, c):
        f, r = s.solve_batch_compact(a, b, c)
        return f.cpu(), r.cpu()                  # gloo gathers host tensors
    front, rear = solve_sharded(solve, v, dy, dphi)
g = np.load(os.path.join({root!r}, "tests", "golden", "compact_H20.npz"))
bits = lambda a: np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)
assert np.array_equal(bits(front.numpy()[:1024]), bits(g["front"])), "front"
assert np.array_equal(bits(rear.numpy()[:1024]), bits(g["rear"])), "rear"
# the other rank's block must have arrived too: compare with a local solve of the whole batch
with MpcSolver(horizon=H, device=0, algo="lane") as s:
    ff, rr = s.solve_batch_compact(v, dy, dphi)
assert np.array_equal(bits(front.numpy()), bits(ff.cpu().numpy())) and np.array_equal(bits(rear.numpy()), bits(rr.cpu().numpy()))
# a batch that arrives sorted by speed: the block split gives one rank the long instances, the interleaved one deals them
# round (SURVEY.md section 8e); same bits in instance order under either, within one family
o = torch.argsort(v)
vs, ys, ps = v[o].contiguous(), dy[o].contiguous(), dphi[o].contiguous()
with MpcSolver(horizon=H, device=0, algo="lane_fma") as s:
    whole = s.solve_batch_compact(vs, ys, ps)
    for split in ("block", "interleaved"):
        tot = []
        def solve2(a, b, c):
            f, r, it = s.solve_batch_compact(a, b, c, want_iters=True)
            tot.append(int(it.sum()))
            return f.cpu(), r.cpu()
        f2, r2 = solve_sharded(solve2, vs, ys, ps, split=split)
        assert np.array_equal(bits(f2.numpy()), bits(whole[0].cpu().numpy())) and np.array_equal(bits(r2.numpy()), bits(whole[1].cpu().numpy())), split
        t = torch.tensor([tot[0]], dtype=torch.int64)
        both = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(both, t)
        ratio = max(int(b) for b in both) / min(int(b) for b in both)
        assert (ratio > 2.0) if split == "block" else (ratio < 1.05), (split, ratio)
print("rank", rank, "of", world, "ok", shard_range(n, rank, world))
open(os.path.join(os.path.dirname(os.path.abspath(__file__)), f"rank{{rank}}.ok"), "w").write("ok")   # (the two ranks' prints can interleave)
dist.destroy_process_group()
'''


def test_product_solver_under_two_rank_sharding(torch_cuda, tmp_path):
    """The PRODUCT solver (not the oracle) under solve_sharded with two torch.distributed ranks:
    a fresh `torch.distributed.run` child (gloo collectives, both ranks on cuda:0 because the test box
    has one card), ragged shards, every rank ends with the full batch; first 1024 = real-dlib fixtures."""
    script = tmp_path / "rank_worker.py"
    script.write_text(_RANK_WORKER.format(root=ROOT))
    # loopback only: the box's hostname may not resolve, and gloo would otherwise pick its address by name
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1", HSA_ENABLE_IPC_MODE_LEGACY="0",
               GLOO_SOCKET_IFNAME="lo")
    import socket
    with socket.socket() as sk:           # a port nobody holds right now
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)],
                         env=env, capture_output=True, text=True, timeout=600)
    if out.returncode != 0:               # keep the evidence where gpurun brings it home
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "two_rank_failure.log"), "w") as fh:
            fh.write(out.stdout + "\n---- stderr ----\n" + out.stderr)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert (tmp_path / "rank0.ok").exists() and (tmp_path / "rank1.ok").exists(), out.stdout[-2000:]


# ---------------------------------------------------------------------------------------------
# one trajectory point per horizon step

def _random_trajectories(n, P, seed):
    rng = np.random.default_rng(seed)
    seg = rng.uniform(0.02, 0.25, size=(P, n)).astype(np.float32)
    ang = np.cumsum(rng.uniform(-0.15, 0.15, size=(P, n)), axis=0).astype(np.float32)
    px = np.cumsum(seg * np.cos(ang), axis=0, dtype=np.float32)
    py = (np.cumsum(seg * np.sin(ang), axis=0, dtype=np.float32) + rng.uniform(-0.2, 0.2, size=n).astype(np.float32))
    dx, dy = np.cos(ang).astype(np.float32), np.sin(ang).astype(np.float32)
    vel = rng.uniform(0.0, 2.0, size=(P, n)).astype(np.float32)
    count = rng.integers(0, P + 1, size=n).astype(np.int32)
    count[:3] = (0, 1, 2)
    carv = rng.uniform(-0.2, 4.0, size=n).astype(np.float32)
    look = rng.uniform(0.2, 1.5, size=n).astype(np.float32)
    return px, py, dx, dy, vel, count, carv, look


def _walk_np(px, py, dx, dy, vel, count, first, spacing, H):
    """float32 restatement of traj_horizon_kernel's walk for one polyline -- test infrastructure.
    Returns per step (ox, oy, odx, ody, ovel)."""
    f = np.float32
    out = []
    t = 0
    want = lambda t: f(first) if t == 0 else f(f(first) + f(f(t) * f(spacing)))
    if count > 0:
        walked = f(0)
        for i in range(1, count):
            if t >= H:
                break
            ex, ey = f(px[i - 1] - px[i]), f(py[i - 1] - py[i])
            ln = f(np.sqrt(f(f(ex * ex) + f(ey * ey))))
            walked = f(walked + ln)
            while t < H and walked > want(t):
                back = f(walked - want(t))
                nx, ny = (f(ex / ln), f(ey / ln)) if ln > 0 else (f(0), f(0))
                out.append((f(px[i] + f(nx * back)), f(py[i] + f(ny * back)), dx[i], dy[i], vel[i]))
                t += 1
        j = count - 1
        while t < H:
            out.append((px[j], py[j], dx[j], dy[j], vel[j]))
            t += 1
    else:
        while t < H:
            out.append((want(t), f(0), f(1), f(0), f(0)))
            t += 1
    return out


def test_follow_batch_horizon(torch_cuda, oracle):
    """(1) spacing 0 puts every horizon step on tpc_mpc_follow_batch's single point: the general-form
    solve must then equal the compact one bit for bit (both are dlib's arithmetic).  (2) With real
    spacings the targets equal a float32 numpy restatement of the walk, and the steering equals the
    oracle's general-form solve of those targets."""
    torch = torch_cuda
    H, n, P = 10, 6000, 24
    px, py, dx, dy, vel, count, carv, look = _random_trajectories(n, P, 5)
    g = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    dev = [g(a) for a in (px, py, dx, dy, vel, count, carv, look)]
    with _solver(H, "lane") as s:
        f1, r1, ts1, td1, it1 = s.follow_batch(*dev, want_iters=True)
        f0, r0, ts0, td0, it0 = s.follow_batch_horizon(*dev, step_spacing=torch.zeros(n, dtype=torch.float32, device="cuda"),
                                                       want_iters=True)
        f2, r2, ts2, td2, tg, it2 = s.follow_batch_horizon(*dev, want_targets=True, want_iters=True)
        torch.cuda.synchronize()
    assert torch.equal(ts0, ts1) and torch.equal(td0, td1) and torch.equal(it0, it1)
    assert bits_equal(f0.cpu().numpy(), f1.cpu().numpy()) and bits_equal(r0.cpu().numpy(), r1.cpu().numpy())
    assert torch.equal(ts2, ts1) and torch.equal(td2, td1)          # step 0 is the same point
    tg = tg.cpu().numpy()
    # expected targets + models
    T, l = 0.1, 0.21
    et = np.zeros((n, H, 2))
    ev = np.zeros(n)
    for k in range(n):
        vv = np.float64(carv[k])
        if abs(vv) < 0.1:
            vv = 0.1
        vv = np.float64(np.float32(vv))                              # the (absent) lookup table works in float
        ev[k] = vv
        spacing = np.float32(abs(vv) * T)
        for t, (ox, oy, odx, ody, ovel) in enumerate(_walk_np(px[:, k], py[:, k], dx[:, k], dy[:, k], vel[:, k],
                                                                  int(count[k]), look[k], spacing, H)):
            et[k, t, 0], et[k, t, 1] = np.float64(oy), np.arctan2(np.float64(ody), np.float64(odx))
    got = tg.T.reshape(n, H, 2)
    assert np.array_equal(got[:, :, 0], et[:, :, 0])                 # y: exact float32 arithmetic
    assert np.abs(got[:, :, 1] - et[:, :, 1]).max() <= 1e-15         # atan2: device vs glibc, last bit
    amax = 22 * np.pi / 180
    A = np.stack([np.ones(n), T * ev, np.zeros(n), np.ones(n)], 1)
    B = np.stack([np.zeros(n), T * ev, T * ev / l, -(T * ev / l)], 1)
    z2 = np.zeros((n, 2))
    u0, _, oit = oracle.solve_general(2, H, A, B, z2, np.tile([20.0, 7.0], (n, 1)), np.tile([0.0005, 10.0], (n, 1)),
                                      np.full((n, 2), -amax), np.full((n, 2), amax), z2, got, nthreads=8)
    crossing = ts1.cpu().numpy() < 0.5
    u0[crossing] = 0.0
    assert np.array_equal(it2.cpu().numpy(), oit)
    assert bits_equal(f2.cpu().numpy(), u0[:, 0]) and bits_equal(r2.cpu().numpy(), u0[:, 1])
    assert np.abs(f2.cpu().numpy() - f1.cpu().numpy()).max() > 1e-3  # per-step targets do change the answer


# ---------------------------------------------------------------------------------------------
# solve_one: the resident kernel

@pytest.mark.parametrize("H", [4, 5, 10, 20, 30, 40])
def test_solve_one_resident_vs_oracle(torch_cuda, oracle, H):
    """tpc_mpc_solve_one through the resident wavefront (every specialised horizon; two variables per lane at H = 40),
    a few dozen different requests in a row, against the oracle; then the same with the resident mode
    off, after an idle timeout (the wave has left and must be started again), and with changed knobs."""
    from trajectory_controller_amd.synth import compact_inputs
    n = 40 if H <= 20 else 12
    v, dy, dphi = compact_inputs(H, n, first=4321)
    of, orr, _ = oracle.solve_compact(H, v, dy, dphi)
    with _solver(H, "auto") as s:
        for k in range(n):
            f, r = s.solve_one(v[k], dy[k], dphi[k])
            assert abs(f - of[k]) <= 1e-9 and abs(r - orr[k]) <= 1e-9, (H, k)
        s.set_resident(1500)                 # 1.5 ms idle timeout
        f, r = s.solve_one(v[0], dy[0], dphi[0])
        time.sleep(0.05)                     # the wave has timed out by now
        for k in range(3):
            f, r = s.solve_one(v[k], dy[k], dphi[k])
            assert abs(f - of[k]) <= 1e-9 and abs(r - orr[k]) <= 1e-9, (H, k)
        s.set_resident(0)                    # ordinary launches
        f, r = s.solve_one(v[1], dy[1], dphi[1])
        assert abs(f - of[1]) <= 1e-9 and abs(r - orr[1]) <= 1e-9
        s.set_resident(20000)
        e2f, e2r, _ = oracle.solve_compact(H, v[:2], dy[:2], dphi[:2], eps=1e-4, max_iter=400, lo=(-0.1, -0.2), hi=(0.3, 0.1))
        f, r = s.solve_one(v[1], dy[1], dphi[1], eps=1e-4, max_iter=400, lower=(-0.1, -0.2), upper=(0.3, 0.1))
        assert abs(f - e2f[1]) <= 1e-9 and abs(r - e2r[1]) <= 1e-9


@pytest.mark.parametrize("H", [4, 10, 20])
def test_solve_one_kept_setup(torch_cuda, oracle, H):
    """The resident wave keeps the model-only part of its set-up (Hessian row, Q_diag, lambda, 1/lambda, beta) while a
    request repeats the previous one's v and parameters: runs of requests with the speed held and the targets moving,
    then a changed speed, changed weights, changed bounds and a changed eps in between -- every answer against the
    oracle for ITS parameters."""
    rng = np.random.default_rng(77 + H)
    with _solver(H, "auto") as s:
        for block in range(6):
            v = float(rng.uniform(0.3, 3.8))
            kw = {}
            if block == 2: kw = dict(weight_y=35.0, weight_steering_rear=4.0)
            if block == 3: kw = dict(lower=(-0.2, -0.3), upper=(0.25, 0.3))
            if block == 4: kw = dict(eps=1e-3)
            okw = {}
            if block == 2: okw = dict(weights=(35.0, 7.0, 0.0005, 4.0))
            if block == 3: okw = dict(lo=(-0.2, -0.3), hi=(0.25, 0.3))
            if block == 4: okw = dict(eps=1e-3)
            for rep in range(6):
                dy, dphi = float(rng.uniform(-0.5, 0.5)), float(rng.uniform(-0.6, 0.6))
                of, orr, _ = oracle.solve_compact(H, [v], [dy], [dphi], **okw)
                f, r = s.solve_one(v, dy, dphi, **kw)
                assert abs(f - of[0]) <= 1e-9 and abs(r - orr[0]) <= 1e-9, (H, block, rep)


def _solve_one_paths(s, capi, path):
    """Put solver `s` on one of tpc_mpc_solve_one's three paths."""
    if path == "launch":          # resident mode off: one WAVE launch per call
        s.set_resident(0)
    elif path == "resident_host":  # request lines in the pinned block instead of device memory behind the BAR
        s.set_option(capi.OPT_MAILBOX_HOST, 1)


@pytest.mark.parametrize("path", ["resident", "resident_host", "launch", "lane"])
@pytest.mark.parametrize("H", [4, 20])
def test_solve_one_edge_cases_and_flags(torch_cuda, oracle, H, path):
    """Every row of the real-dlib edge fixture (NaN v / delta_y / delta_phi, zero target, saturating targets, v = 1e-3
    and 50) through the call that replaces mpcControllerTobi, on each of its paths -- the resident wavefront (request
    lines behind the BAR or in the pinned block), one WAVE launch per call, one LANE launch per call: dlib's outputs
    (a NaN input returns the untouched start point, mpc.h:298-311), the non-fatal flags and the iteration count
    through tpc_mpc_last_flags, and no call anywhere near the resident path's 2 s give-up."""
    from trajectory_controller_amd import FLAG_MAX_ITER, FLAG_NONFINITE, capi
    g = load_golden("compact_edge.npz")
    _, _, oit = oracle.solve_compact(H, g["v"], g["dy"], g["dphi"])
    algo = "lane" if path == "lane" else "auto"
    with _solver(H, algo) as s:
        _solve_one_paths(s, capi, path)
        s.solve_one(1.0, 0.1, 0.05)   # (start-up of the resident wave is not what is timed below)
        for rep in range(2):
            for k in range(len(g["v"])):
                t0 = time.perf_counter()
                f, r = s.solve_one(g["v"][k], g["dy"][k], g["dphi"][k])
                dt = time.perf_counter() - t0
                flags, it = s.last_solve_one_flags()
                ef, er = g[f"front_H{H}"][k], g[f"rear_H{H}"][k]
                if path == "lane":
                    assert bits_equal([f, r], [ef, er]), (k, f, r)
                else:
                    assert abs(f - ef) <= 1e-9 and abs(r - er) <= 1e-9, (k, f, r)
                nan_in = bool(np.isnan(g["v"][k]) or np.isnan(g["dy"][k]) or np.isnan(g["dphi"][k]))
                # (the v = 50 row ends on the iteration cap in dlib too: that is what the second flag says)
                assert flags == (FLAG_NONFINITE if nan_in else (FLAG_MAX_ITER if oit[k] == 10000 else 0)), (k, flags)
                assert it == oit[k], (k, it, oit[k])
                if nan_in:
                    assert f == 0 and r == 0 and it == 0
                assert dt < 0.5, (k, dt)
        # infinities are non-finite inputs too; a finite request right behind them is answered as usual
        for v, dy, dphi in ((np.inf, 0.1, 0.1), (1.0, -np.inf, 0.1), (1.0, 0.1, np.inf), (-np.inf, np.nan, 0.0)):
            f, r = s.solve_one(v, dy, dphi)
            assert f == 0 and r == 0 and s.last_solve_one_flags() == (FLAG_NONFINITE, 0)
        # a solve cut off by max_iter says so, and returns the iterate dlib returns
        of, orr, oit2 = oracle.solve_compact(H, [2.0], [-0.2], [0.1], max_iter=30)
        f, r = s.solve_one(2.0, -0.2, 0.1, max_iter=30)
        assert s.last_solve_one_flags() == (FLAG_MAX_ITER, 30) and oit2[0] == 30
        assert abs(f - of[0]) <= 1e-9 and abs(r - orr[0]) <= 1e-9
        f, r = s.solve_one(1.0, 0.1, 0.05)
        of, orr, oit3 = oracle.solve_compact(H, [1.0], [0.1], [0.05])
        assert abs(f - of[0]) <= 1e-9 and abs(r - orr[0]) <= 1e-9 and s.last_solve_one_flags() == (0, int(oit3[0]))


def test_resident_off_survives_a_mailbox_restart(torch_cuda):
    """tpc_mpc_set_resident(0) -- every solve_one an ordinary launch -- must still hold after
    tpc_mpc_set_option(TPC_MPC_OPT_MAILBOX_HOST), which tears the single-solve state down and sets it up again.
    Observable through the profiling events: only the launch path records kernel times."""
    from trajectory_controller_amd import TpcMpcError, capi
    with _solver(4, "auto") as s:
        s.set_profiling(True)
        s.solve_one(1.0, 0.1, 0.05)                  # resident: no launch, no kernel times
        with pytest.raises(TpcMpcError):
            s.last_kernel_times()
        s.set_resident(0)
        s.set_option(capi.OPT_MAILBOX_HOST, 1)
        f, r = s.solve_one(1.0, 0.1, 0.05)
        assert s.last_kernel_times()[2] == capi.ALGO_WAVE
        assert abs(f - 0.28258865451261717) <= 1e-12 and abs(r - 0.059891817493776013) <= 1e-12


def test_solve_one_host_option_on_a_gpu_handle(torch_cuda, oracle):
    """TPC_MPC_OPT_HOST_SOLVE_ONE: a GPU handle solves single instances up to the named horizon on the calling thread
    (csrc/tpc_mpc_host.cpp) -- the same bits a host-only handle returns, within 1e-9 of the resident wavefront's and of
    the oracle's, longer horizons still go to the GPU, and a solve that ends on the cap under AUTO comes back with
    dlib's bits (the bit-exact LANE kernels)."""
    from trajectory_controller_amd import FLAG_MAX_ITER, MpcSolver, capi
    from trajectory_controller_amd.synth import compact_inputs
    with _solver(4, "auto") as g, MpcSolver(horizon=4, device=capi.DEVICE_NONE) as hst, _solver(4, "auto") as res:
        g.set_option(capi.OPT_HOST_SOLVE_ONE, 10)
        g.set_profiling(True)
        for H in (4, 10):
            v, dy, dphi = compact_inputs(H, 60, first=5150 + H)
            of, orr, oit = oracle.solve_compact(H, v, dy, dphi)
            for k in range(60):
                a = g.solve_one(v[k], dy[k], dphi[k], horizon=H)
                assert a == hst.solve_one(v[k], dy[k], dphi[k], horizon=H)
                assert g.last_solve_one_flags() == (0, int(oit[k]))
                b = res.solve_one(v[k], dy[k], dphi[k], horizon=H)
                assert abs(a[0] - b[0]) <= 1e-9 and abs(a[1] - b[1]) <= 1e-9
                assert abs(a[0] - of[k]) <= 1e-9 and abs(a[1] - orr[k]) <= 1e-9
        # beyond the option's horizon: the GPU as before
        f, r = g.solve_one(2.0, -0.2, 0.1, horizon=20)
        assert abs(f + 0.34544297733739515) <= 1e-12 and abs(r + 0.21765810887303699) <= 1e-12
        # on the cap under AUTO: dlib's bits, through one LANE launch (the only launch this handle has made)
        of, orr, oit = oracle.solve_compact(4, [50.0], [0.3], [0.2])
        assert oit[0] == 10000
        f, r = g.solve_one(50.0, 0.3, 0.2, horizon=4)
        assert bits_equal([f, r], [of[0], orr[0]]) and g.last_solve_one_flags() == (FLAG_MAX_ITER, 10000)
        assert g.last_kernel_times()[2] == capi.ALGO_LANE


def test_last_flags_before_any_solve_one(torch_cuda):
    from trajectory_controller_amd import TpcMpcError
    with _solver(4, "auto") as s:
        with pytest.raises(TpcMpcError):
            s.last_solve_one_flags()


@pytest.mark.parametrize("where", ["device", "host"])
def test_solve_one_request_lines_placement(torch_cuda, oracle, where):
    """The resident wave takes its requests from device memory the CPU writes through the BAR (where the
    part has a large BAR) or from the pinned block (TPC_MPC_OPT_MAILBOX_HOST, the fallback): same answers, also
    across a horizon swap, an idle timeout and two handles alive at once."""
    from trajectory_controller_amd import capi
    from trajectory_controller_amd.synth import compact_inputs
    ref = {}
    for H in (4, 10):
        v, dy, dphi = compact_inputs(H, 24, first=991 + H)
        ref[H] = (v, dy, dphi) + tuple(oracle.solve_compact(H, v, dy, dphi)[:2])
    with _solver(4, "auto") as s, _solver(10, "auto") as t:      # the placement is chosen per handle
        if where == "host":
            s.set_option(capi.OPT_MAILBOX_HOST, 1)
            t.set_option(capi.OPT_MAILBOX_HOST, 1)
        for rep in range(2):
            for k in range(24):
                for slv, H in ((s, 4), (t, 10), (s, 10)):        # s swaps its resident wave between two horizons
                    v, dy, dphi, of, orr = ref[H]
                    f, r = slv.solve_one(v[k], dy[k], dphi[k], horizon=H) if slv is s else slv.solve_one(v[k], dy[k], dphi[k])
                    assert abs(f - of[k]) <= 1e-9 and abs(r - orr[k]) <= 1e-9, (where, H, k)
            s.set_resident(1000)
            time.sleep(0.03)                                     # s's wave has left: the next request restarts it


def test_solve_one_latency_and_coexistence(torch_cuda):
    """Latency of the resident path at the reference's real size (H = 4), printed; and a batch solve on
    the same device while the resident wave is up (they must not disturb each other)."""
    from trajectory_controller_amd.synth import compact_inputs
    torch = torch_cuda
    with _solver(4, "auto") as s, _solver(20, "lane") as b:
        s.solve_one(1.0, 0.1, 0.05)
        t0 = time.perf_counter()
        for k in range(2000):
            s.solve_one(1.0 + 1e-4 * k, 0.1, 0.05)
        dt = (time.perf_counter() - t0) / 2000
        print(f"solve_one resident, H=4: {dt * 1e6:.2f} us per call")
        v, dy, dphi = (torch.from_numpy(a).cuda() for a in compact_inputs(20, 70000))
        f, r = b.solve_batch_compact(v, dy, dphi)
        f1, r1 = s.solve_one(1.0, 0.1, 0.05)
        g = load_golden("compact_H20.npz")
        assert bits_equal(f[:1024].cpu().numpy(), g["front"])
        assert abs(f1 - 0.28258865451261717) <= 1e-12 and abs(r1 - 0.059891817493776013) <= 1e-12
    assert dt < 200e-6


def _build_example(tmp_path, name, extra=()):
    import subprocess
    lib = os.path.join(ROOT, "trajectory_controller_amd", "lib")
    exe = str(tmp_path / name)
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-D__HIP_PLATFORM_AMD__", "-I" + os.path.join(ROOT, "include"),
                           "-I/opt/rocm/include", os.path.join(ROOT, "examples", name + ".c"), "-o", exe, "-L" + lib,
                           "-ltpc_mpc", "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath," + lib, *extra])
    return exe


def test_sharded_inprocess_example(torch_cuda, tmp_path):
    """examples/sharded_inprocess.c = INTEGRATION.md section 3 compiled as C: one host process, one handle and
    stream per GPU, one RCCL communicator, the sharded call per GPU between group_begin / group_end.  On the
    one-GPU test box G = 1 and the example asks for a real one-rank communicator, so RCCL's all-gather runs;
    it verifies by itself that every GPU holds all outputs, bit-identical to a one-GPU solve."""
    import json, subprocess
    exe = _build_example(tmp_path, "sharded_inprocess")
    r = subprocess.run([exe, "0", "20011", "10"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    last = json.loads(r.stdout.strip().splitlines()[-1])
    assert last["verified"] is True and last["gpus"] >= 1 and last["n_total"] == 20011
