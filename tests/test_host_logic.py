"""CPU-only tests of the host side: the C-ABI library loads and exports every symbol the header
declares (no compute calls: there is no GPU here), the synthetic-input generator is reproducible,
and the multi-GPU sharding path works at world_size 2 over gloo."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    text = open(os.path.join(ROOT, "include", "tpc_mpc.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(tpc_mpc_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_whole_abi():
    from trajectory_controller_amd import capi
    decl = _header_functions()
    assert len(decl) >= 10
    assert sorted(capi.EXPORTS) == decl, "capi.EXPORTS must list exactly what include/tpc_mpc.h declares"
    lib = capi.load_library()          # raises if the .so is missing or lacks a symbol
    out = subprocess.check_output(["nm", "-D", "--defined-only", capi.LIB_PATH], text=True)
    exported = set(re.findall(r" T (tpc_mpc_[a-z_0-9]+)", out))
    assert set(decl) <= exported
    assert lib.tpc_mpc_abi_version() == capi.ABI_VERSION == 5


def test_header_is_plain_c(tmp_path):
    """The boundary is a C ABI: the header must compile as C99 (and as C++) on its own, and a C
    translation unit that calls every entry point must link against the library."""
    hdr = os.path.join(ROOT, "include")
    src = tmp_path / "abi_link.c"
    calls = "\n".join(f"    use((fn_t){name});" for name in _header_functions())
    src.write_text('#include "tpc_mpc.h"\ntypedef void (*fn_t)(void);\nstatic void use(fn_t f) { (void)f; }\n'
                   'int main(void) {\n' + calls + '\n    return tpc_mpc_abi_version() == TPC_MPC_ABI_VERSION ? 0 : 1;\n}\n')
    from trajectory_controller_amd import capi
    libdir = os.path.dirname(capi.LIB_PATH)
    exe = str(tmp_path / "abi_link")
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-I" + hdr, str(src), "-o", exe,
                           "-L" + libdir, "-ltpc_mpc", "-Wl,-rpath," + libdir])
    subprocess.check_call(["g++", "-std=c++11", "-fsyntax-only", "-I" + hdr, "-x", "c++",
                           os.path.join(hdr, "tpc_mpc.h")])
    assert subprocess.run([exe]).returncode == 0


def test_default_params_match_reference_defaults():
    from trajectory_controller_amd import capi
    p = capi.default_params(4)
    assert (p.horizon, p.eps, p.max_iter, p.smo_iters) == (4, 0.01, 10000, 50)           # mpc.h:103-104,319
    assert (p.step_size, p.wheelbase) == (0.1, 0.21)                                     # follower.cpp:96, .h:47
    assert (p.weight_y, p.weight_phi, p.weight_steering_front, p.weight_steering_rear) == (20, 7, 0.0005, 10)
    assert p.upper[0] == 22 * np.pi / 180 and p.lower[1] == -22 * np.pi / 180           # follower.cpp:16-18
    assert capi.default_params(7).horizon == 7            # any 1..64: generic kernel
    with pytest.raises(capi.TpcMpcError):
        capi.default_params(65)
    import ctypes as C
    hs = (C.c_int * 16)()
    n = capi.load_library().tpc_mpc_supported_horizons(hs, 16)
    assert list(hs[:n]) == [4, 5, 10, 20, 30, 40]


def test_product_never_imports_oracle():
    """The checker must not leak into the product path (task section 3)."""
    pkg = os.path.join(ROOT, "trajectory_controller_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", ".hpp")) or f == "Makefile":
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                for line in text.splitlines():
                    s = line.strip()
                    if s.startswith(("#", "//", "*", '"""')) and "include" not in s:
                        continue
                    assert not re.search(r"(from|import)\s+oracle\b", s), (f, s)
                    assert "liboracle" not in s and "mpc_oracle" not in s, (f, s)
                    # ... nor the CPU model of the LANE_FMA arithmetic (tests/model/): it includes the product's
                    # header, never the other way round
                    assert not re.search(r"(from|import)\s+tests\b", s), (f, s)
                    assert "libub_model" not in s and "ub_model.cpp" not in s and "ub_model_solve" not in s, (f, s)


def test_no_gpu_means_loud_failure():
    """Without a usable gfx950 device the product refuses to run (no CPU fallback)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from trajectory_controller_amd import MpcSolver, TpcMpcError
    with pytest.raises(TpcMpcError) as e:
        MpcSolver(horizon=10)
    assert e.value.status == 6


def test_synth_reproducible_and_in_range():
    from trajectory_controller_amd.synth import compact_inputs, splitmix64_uniform
    # splitmix64 known answers for seed 0 (first outputs 0xE220A8397B1DCDAF, 0x6E789E6AA1B965F4)
    u = splitmix64_uniform(0, 2)
    assert u[0] == (0xE220A8397B1DCDAF >> 11) / 2.0**53 and u[1] == (0x6E789E6AA1B965F4 >> 11) / 2.0**53
    v, dy, dphi = compact_inputs(20, 1000)
    v2, dy2, dphi2 = compact_inputs(20, 300, first=500)
    assert np.array_equal(v[500:800], v2) and np.array_equal(dy[500:800], dy2) and np.array_equal(dphi[500:800], dphi2)
    assert v.min() >= 0.1 and v.max() <= 4.0 and np.abs(dy).max() <= 0.5 and np.abs(dphi).max() <= 0.6
    g = np.load(os.path.join(ROOT, "tests", "golden", "compact_H20.npz"))
    assert np.array_equal(g["v"], v[:0] if False else compact_inputs(20, 1024)[0])


def test_shard_ranges_cover():
    from trajectory_controller_amd.shard import shard_range
    for n in (0, 1, 7, 64, 1000, 262144):
        for world in (1, 2, 3, 8):
            blocks = [shard_range(n, r, world) for r in range(world)]
            assert blocks[0][0] == 0 and sum(c for _, c in blocks) == n
            for (f0, c0), (f1, _) in zip(blocks, blocks[1:]):
                assert f0 + c0 == f1
            assert max(c for _, c in blocks) - min(c for _, c in blocks) <= 1


def test_shard_maps_cover():
    """tpc_mpc_shard_map (C ABI) and shard.shard_slice agree, and each split's shards partition the batch."""
    import ctypes as C
    from trajectory_controller_amd import capi
    from trajectory_controller_amd.shard import shard_slice
    lib = capi.load_library()
    for n in (0, 1, 7, 64, 1001, 4096):
        for world in (1, 2, 3, 5, 8):
            for name, split in capi.SPLITS.items():
                seen = np.zeros(n, dtype=np.int32)
                for rank in range(world):
                    first, count, stride = C.c_int64(), C.c_int64(), C.c_int64()
                    assert lib.tpc_mpc_shard_map(n, rank, world, split, C.byref(first), C.byref(count), C.byref(stride)) == capi.OK
                    idx = first.value + stride.value * np.arange(count.value)
                    assert np.array_equal(idx, np.arange(n)[shard_slice(n, rank, world, name)])
                    seen[idx] += 1
                assert np.all(seen == 1)
    assert lib.tpc_mpc_shard_map(10, 2, 2, 0, C.byref(first), C.byref(count), C.byref(stride)) == 1   # TPC_MPC_ERR_BAD_ARG
    assert lib.tpc_mpc_shard_map(10, 0, 2, 7, C.byref(first), C.byref(count), C.byref(stride)) == 1   # TPC_MPC_ERR_BAD_ARG


def test_exchange_plan_on_a_host_communicator():
    """The slot arithmetic of the sharded entries for 2..8 owners, without a GPU: tpc_mpc_x_exchange_plan hands out the very
    list of collectives tpc_mpc_solve_batch_compact_sharded_split / tpc_mpc_gather_shards_split issue for one output row
    (the same function feeds their RCCL calls); here a host-memory communicator executes it for every rank at once --
    equal blocks (in-place all-gather), ragged blocks (one in-place broadcast per owner, also forced on sizes that divide),
    and the interleaved split's staged all-gather followed by the un-permutation."""
    import ctypes as C
    from trajectory_controller_amd import capi
    lib = capi.load_library()
    for world in range(2, 9):
        for n in (1, world - 1, world, 7 * world, 7 * world + 3, 1001, 4096):
            for split in (capi.SPLIT_BLOCK, capi.SPLIT_INTERLEAVED):
                for ragged in (0, 1):
                    plans, bufs = [], []
                    for rank in range(world):
                        ops = (C.c_int64 * (5 * 64))()
                        blen = C.c_int64()
                        k = lib.tpc_mpc_x_exchange_plan(n, world, rank, split, ragged, ops, 64, C.byref(blen))
                        assert 0 < k <= 64
                        plans.append(np.array(ops[:5 * k], dtype=np.int64).reshape(k, 5))
                        first, count, stride = C.c_int64(), C.c_int64(), C.c_int64()
                        lib.tpc_mpc_shard_map(n, rank, world, split, C.byref(first), C.byref(count), C.byref(stride))
                        mine = first.value + stride.value * np.arange(count.value)     # the instances this rank solved
                        buf = np.full(blen.value, -1, dtype=np.int64)
                        if split == capi.SPLIT_BLOCK:
                            buf[mine] = mine                                            # its slot of the full-size row
                        else:
                            cap = blen.value // world
                            buf[rank * cap: rank * cap + count.value] = mine            # its slot of the staging array
                        bufs.append(buf)
                    assert all(len(pl) == len(plans[0]) for pl in plans)               # a collective needs every rank
                    for i in range(len(plans[0])):
                        kind = plans[0][i, 0]
                        assert all(pl[i, 0] == kind for pl in plans)
                        if kind == 0:       # all-gather: rank q's `count` elements land at recv_off + q * count everywhere
                            cnt = plans[0][i, 4]
                            assert all(pl[i, 4] == cnt and pl[i, 3] == plans[0][i, 3] for pl in plans)
                            parts = [bufs[q][plans[q][i, 2]: plans[q][i, 2] + cnt].copy() for q in range(world)]
                            for q in range(world):
                                for src in range(world):
                                    ro = plans[q][i, 3] + src * cnt
                                    bufs[q][ro: ro + cnt] = parts[src]
                        else:               # broadcast from root, in place
                            root, so, ro, cnt = plans[0][i, 1:5]
                            assert all(tuple(pl[i, 1:5]) == (root, so, ro, cnt) for pl in plans)
                            data = bufs[root][so: so + cnt].copy()
                            for q in range(world):
                                bufs[q][ro: ro + cnt] = data
                    for q in range(world):
                        if split == capi.SPLIT_BLOCK:
                            out = bufs[q]
                        else:               # unpermute_kernel: out[i] = stage[(i mod world) * cap + i // world]
                            cap = len(bufs[q]) // world
                            i_ = np.arange(n)
                            out = bufs[q][(i_ % world) * cap + i_ // world]
                        assert np.array_equal(out, np.arange(n)), (world, n, split, ragged, q)


_GLOO_WORKER = r'''
import os, sys
sys.path.insert(0, {root!r})
import numpy as np, torch, torch.distributed as dist
from oracle.bindings import Oracle            # the checker stands in for the GPU solve here (tests only)
from trajectory_controller_amd.shard import solve_sharded
from trajectory_controller_amd.synth import compact_inputs
dist.init_process_group("gloo")
H, n = 5, 1001                                 # ragged: 501 + 500
v, dy, dphi = (torch.from_numpy(a) for a in compact_inputs(H, n))
orc = Oracle()
def solve(a, b, c):
    f, r, _ = orc.solve_compact(H, a.numpy(), b.numpy(), c.numpy())
    return torch.from_numpy(f), torch.from_numpy(r)
front, rear = solve_sharded(solve, v, dy, dphi)
ef, er, _ = orc.solve_compact(H, v.numpy(), dy.numpy(), dphi.numpy())
assert np.array_equal(front.numpy(), ef) and np.array_equal(rear.numpy(), er)
for turn in range(2):
    dist.barrier()
    if turn == dist.get_rank():
        sys.stdout.write("rank %d ok\n" % turn); sys.stdout.flush()
dist.barrier()
dist.destroy_process_group()
'''


def test_sharded_solve_world2_gloo(tmp_path):
    """N>1 path on CPU: two ranks, ragged shards, all-gather over gloo, full result on every rank."""
    script = tmp_path / "worker.py"
    script.write_text(_GLOO_WORKER.format(root=ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1", GLOO_SOCKET_IFNAME="lo")
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.count("ok") == 2


_GLOO_SKEW_WORKER = r'''
import os, sys
sys.path.insert(0, {root!r})
import numpy as np, torch, torch.distributed as dist
from oracle.bindings import Oracle            # the checker stands in for the GPU solve here (tests only)
from trajectory_controller_amd.shard import solve_sharded
from trajectory_controller_amd.synth import compact_inputs
dist.init_process_group("gloo")
H, n = 10, 2001
v, dy, dphi = compact_inputs(H, n)
o = np.argsort(v)                              # a host that hands over its batch SORTED BY SPEED
v, dy, dphi = (torch.from_numpy(np.ascontiguousarray(a[o])) for a in (v, dy, dphi))
orc = Oracle()
ef, er, _ = orc.solve_compact(H, v.numpy(), dy.numpy(), dphi.numpy())
work = {{}}
for split in ("block", "interleaved"):
    def solve(a, b, c):
        f, r, it = orc.solve_compact(H, a.numpy(), b.numpy(), c.numpy())
        work[split] = int(it.sum())            # the rank's iteration total = its solve time on a GPU
        return torch.from_numpy(f), torch.from_numpy(r)
    front, rear = solve_sharded(solve, v, dy, dphi, split=split)
    assert np.array_equal(front.numpy(), ef) and np.array_equal(rear.numpy(), er), split   # instance order, same bits
    t = torch.tensor([work[split]], dtype=torch.int64)
    both = [torch.zeros_like(t) for _ in range(2)]
    dist.all_gather(both, t)
    a, b = int(both[0]), int(both[1])
    ratio = max(a, b) / min(a, b)
    if dist.get_rank() == 0:   # (one write per line, and the other rank silent until the barrier below: the launcher interleaves the ranks' output)
        sys.stdout.write("%s iteration totals per rank %d %d max/min %.3f\n" % (split, a, b, ratio)); sys.stdout.flush()
    if split == "block":
        assert ratio > 2.0, ratio              # one rank gets the slow half
    else:
        assert ratio < 1.05, ratio
for turn in range(2):
    dist.barrier()
    if turn == dist.get_rank():
        sys.stdout.write("rank %d ok\n" % turn); sys.stdout.flush()
dist.barrier()
dist.destroy_process_group()
'''


def test_interleaved_split_balances_a_speed_sorted_batch_world2_gloo(tmp_path):
    """the iteration count is a function of the speed, so a block split of a speed-sorted batch is
    > 2x out of balance; the interleaved split is within 5 %, and both return the same bits in instance order."""
    script = tmp_path / "worker.py"
    script.write_text(_GLOO_SKEW_WORKER.format(root=ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1", GLOO_SOCKET_IFNAME="lo")
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.count("ok") == 2 and "interleaved iteration totals" in out.stdout


def test_examples_compile_as_c99(tmp_path):
    """Every C host under examples/ builds as plain C99 against the header and links the library (they are run
    on the GPU box by tests/test_abi_gpu.py)."""
    from trajectory_controller_amd import capi
    libdir = os.path.dirname(capi.LIB_PATH)
    for name in sorted(os.listdir(os.path.join(ROOT, "examples"))):
        if not name.endswith(".c"):
            continue
        subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-D__HIP_PLATFORM_AMD__",
                               "-I" + os.path.join(ROOT, "include"), "-I/opt/rocm/include",
                               os.path.join(ROOT, "examples", name), "-o", str(tmp_path / name[:-2]), "-L" + libdir,
                               "-ltpc_mpc", "-L/opt/rocm/lib", "-lamdhip64", "-lm", "-Wl,-rpath," + libdir])
