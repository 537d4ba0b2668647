#!/usr/bin/env python3
"""bench.py -- MPC QP solves/sec on MI355X (BASELINE.json metric), one JSON line on rank 0.

    python bench.py [--gpus N --steps K --warmup W]            (N=1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch of synthetic instances per GPU: the compact
(reference call pattern) form, horizon 20, 2 inputs, 262 144 instances per GPU (BASELINE config 3;
at N=8 that is config 4's 2 097 152), inputs resident in HBM before the timed region.  The instance
batch is embarrassingly parallel, so ranks shard it with no data-path collective during the solve;
for N>1 every step is ONE library call per rank, tpc_mpc_solve_batch_compact_sharded: the rank's block
is solved straight into its slot of the full-size output arrays and the slots are all-gathered over
RCCL/xGMI on the same stream, inside the timed region (north_star).  torch.distributed only carries
the 128-byte communicator id to the ranks, the barriers and the max-over-ranks of the elapsed time.

One batch is in flight by default.  `--inflight 2` alternates consecutive steps between two library
handles on two HIP streams, so that the next batch's kernels fill the CUs the previous batch's last
long-running instances no longer need (a batch ends when its slowest lane does): +3 % throughput at
H = 20.  It is not the default because the HIP events that time each kernel then include the time
its workgroups wait for CUs and stop agreeing with rocprofv3's dispatch timestamps.  "kernel_ms" is
the average over the timed region of the library's events around each kernel (last solve of each
handle), "kernel_ms_serial" the same measured right after the timed region with one batch in flight.
A one-GPU run with the default one batch in flight also times the same K steps with two in flight and
reports that figure under "pipelined" (value, ms_per_step, alu_frac), next to `value`.
The default one-GPU run also times BASELINE config 2 (4 096 instances, N = 10, the WAVE family) and config 5
(65 536 instances of mixed horizons through one call) and reports them under "config2" and "config5".

`value` is measured in fp64, the only dtype that meets the 1e-6 parity target (SURVEY.md section 0
fact 3), through AUTO, i.e. the LANE_FMA family (unit-box coordinates, fused multiply-adds: dlib's
iteration on quantities that differ from dlib's by rounding).  Its outputs are compared with real dlib
run on the host in the same run ("max_abs_du_vs_dlib", "max_rel_du_vs_dlib": north_star's bar is 1e-6
relative) and, over the whole batch, with the bit-exact LANE family ("vs_bit_exact": fraction of equal
iteration counts, max |du|).  The bit-exact family's own rate is timed right after and reported under
"bit_exact" (its outputs equal dlib's bit for bit).  The fp32 rate of the same workload is reported
under "fp32" with its error histogram against the fp64 result.  `roofline` prices the dominant kernel against HBM as the
contract asks (this path moves 40 B per solve, so that fraction is tiny by construction), and
"alu" prices the same kernel against the FP64 vector peak with ALGORITHMIC flops
((46H-16) x mean iterations).  `cpu_baseline` is the real dlib path
(oracle/_ref, kind "reference") or, where that build is absent, the C restatement (kind "port"),
timed on this host's cores on a bounded sample; the GPU outputs of that sample are compared with
it and the result is reported as "max_abs_du_vs_dlib".
"""
import argparse
import json
import os
import sys
import time

# dmabuf IPC (the only kind this pool's host driver supports): RCCL between the ranks of one node needs it; the driver's
# environment has it already -- this is for a launch from a bare shell
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_VECTOR_PEAK_TF = 78.6     # MI355X vendor FP64 vector peak
FP32_VECTOR_PEAK_TF = 157.3


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=262144, help="instances per GPU")
    ap.add_argument("--horizon", type=int, default=20)
    ap.add_argument("--algo", default="auto", choices=["auto", "lane", "lane_fma", "wave", "group"])
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32"])
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the CPU baseline leg")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-fp32", action="store_true")
    ap.add_argument("--inflight", type=int, default=1, help="batches kept in flight (handle + stream each)")
    ap.add_argument("--no-pipelined", action="store_true", help="skip the two-batches-in-flight leg")
    ap.add_argument("--no-config2", action="store_true", help="skip the BASELINE config 2 leg (4 096 x N=10, WAVE)")
    ap.add_argument("--no-bit-exact", action="store_true", help="skip the bit-exact LANE family leg")
    ap.add_argument("--no-config5", action="store_true", help="skip the BASELINE config 5 leg (65 536 mixed horizons)")
    ap.add_argument("--no-config1", action="store_true", help="skip the BASELINE config 1 leg (one trajectory, N=10: solve_one latency)")
    ap.add_argument("--no-config4", action="store_true",
                    help="skip the leg that solves BASELINE config 4's whole batch (2 097 152 x N=20, fp32) on this one GPU")
    ap.add_argument("--no-mid", action="store_true", help="skip the mid-size batch leg (16 384 x N=20 through AUTO)")
    ap.add_argument("--split", default="block", choices=["block", "interleaved"],
                    help="N > 1: how the batch is split over the ranks (tpc_mpc_solve_batch_compact_sharded_split)")
    ap.add_argument("--sorted-by-speed", action="store_true",
                    help="the synthetic stream arrives sorted by speed (what the interleaved split is for: the iteration count is a "
                         "function of the speed, a block split then gives one rank all the long instances)")
    ap.add_argument("--allow-fallback-gather", action="store_true",
                    help="N > 1 only: let the run continue with a torch.distributed gather when the library's RCCL path "
                         "cannot be set up (the line then carries \"gather\": \"fallback\"); by default such a run exits non-zero")
    return ap.parse_args()


def usable_cores():
    """Host cores this process may actually use: affinity mask, capped by the cgroup CPU quota."""
    cores = os.cpu_count() or 1
    try:
        cores = min(cores, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    cores = min(cores, max(1, int(float(txt[0]) / float(txt[1]) + 0.5)))
            else:
                q = int(txt[0])
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    cores = min(cores, max(1, int(q / per + 0.5)))
            break
        except Exception:
            continue
    env = os.environ.get("TPC_BENCH_CPU_THREADS")
    return int(env) if env else cores


def library_sha256():
    import hashlib
    from trajectory_controller_amd import capi
    path = os.environ.get("TPC_MPC_LIB", capi.LIB_PATH)
    try:
        return hashlib.sha256(open(path, "rb").read()).hexdigest()
    except OSError:
        return None


def cpu_baseline(H, v, dy, dphi, budget_s):
    """Time the dlib CPU path on this host's cores on a bounded sample of the workload.
    Returns (dict for the JSON line, sample size, reference outputs)."""
    from oracle import bindings as ob
    cores = usable_cores()
    if os.path.exists(ob.REF_SO):
        ref, kind = ob.DlibRef(ob.REF_SO), "reference"
        run = lambda n, th: ref.solve_compact(H, v[:n], dy[:n], dphi[:n], nthreads=th)
    else:
        orc, kind = ob.Oracle(), "port"
        run = lambda n, th: orc.solve_compact(H, v[:n], dy[:n], dphi[:n], nthreads=th)[:2]
    # probe one thread to size the sample for ~budget_s of wall time on all cores
    probe = 128
    t0 = time.perf_counter()
    run(probe, 1)
    per = (time.perf_counter() - t0) / probe
    n = int(min(len(v), max(cores * 64, budget_s * cores / per)))
    n -= n % 64
    t0 = time.perf_counter()
    f, r = run(n, cores)
    dt = time.perf_counter() - t0
    return ({"value": n / dt, "unit": "solves/s", "cores": cores, "kind": kind,
             "per_core": n / dt / cores,
             "sample": f"first {n} instances of the workload (H={H}, fp64, cold start), "
                       f"{cores} threads, {dt:.1f} s"}, n, f, r)


def main():
    a = parse()
    import torch
    import torch.distributed as dist
    from trajectory_controller_amd import MpcSolver
    from trajectory_controller_amd.synth import compact_inputs

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            raise SystemExit("--gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        a.gpus = world
    # rehearsal knobs (not used by the driver): run several ranks on one card with gloo collectives
    backend = os.environ.get("TPC_BENCH_BACKEND", "nccl")
    if "TPC_BENCH_DEVICE" in os.environ:
        local_rank = int(os.environ["TPC_BENCH_DEVICE"])
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)   # RCCL over xGMI
        else:
            dist.init_process_group(backend)

    H, n = a.horizon, a.batch
    # weak scaling: rank r owns its shard of the world * n instances of the horizon-H stream: [r*n, (r+1)*n) under the
    # block split, r, r + world, ... under the interleaved one
    n_total = world * n
    from trajectory_controller_amd.shard import shard_slice
    my = shard_slice(n_total, rank, world, a.split)
    if a.split == "block" and not a.sorted_by_speed:
        v, dy, dphi = compact_inputs(H, n, first=rank * n)
    else:
        v, dy, dphi = compact_inputs(H, n_total)
        if a.sorted_by_speed:
            o = np.argsort(v, kind="stable")
            v, dy, dphi = v[o], dy[o], dphi[o]
        v, dy, dphi = (np.ascontiguousarray(x[my]) for x in (v, dy, dphi))
    tdt = torch.float64 if a.dtype == "f64" else torch.float32
    tv, ty, tp = (torch.from_numpy(x).to(dev, dtype=tdt) for x in (v, dy, dphi))
    slots = max(1, a.inflight)
    fronts = [torch.empty_like(tv) for _ in range(slots)]
    rears = [torch.empty_like(tv) for _ in range(slots)]
    iters_t = None
    if world > 1:
        # full-size outputs: every rank ends each step holding the controls of all world*n instances
        front_all = [torch.empty(n_total, dtype=tdt, device=dev) for _ in range(slots)]
        rear_all = [torch.empty(n_total, dtype=tdt, device=dev) for _ in range(slots)]

    solvers = [MpcSolver(horizon=H, device=local_rank, dtype=a.dtype, algo=a.algo) for _ in range(slots)]
    for sv in solvers:
        sv.set_profiling(True)
        sv.reserve(n)   # scratch is allocated here, not by the first solve of each handle
    solver = solvers[0]
    streams = [torch.cuda.Stream(dev) for _ in range(slots)]
    gather_path = "none (1 GPU)"
    if world > 1:
        # The library's own RCCL communicator (one per handle): rank 0 makes the id, torch.distributed
        # hands it to the other ranks.  If the library cannot set it up (RCCL not loadable), say so
        # loudly and gather with torch.distributed instead -- the solve is the library's either way.
        # Step 1, local and therefore safe to fail on one rank only: can this rank load RCCL and make an id?
        # Step 2 only if EVERY rank can: ncclCommInitRank is a collective, a rank that stayed away from
        # it would leave the others waiting.
        probe_err = ""
        try:
            my_id = MpcSolver.comm_unique_id()
        except Exception as exc:   # noqa: BLE001 -- reported, not hidden
            my_id, probe_err = bytes(128), str(exc)
        ok = torch.tensor([0 if probe_err else 1], device=dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 1:
            try:
                for sv in solvers:
                    idt = torch.frombuffer(bytearray(MpcSolver.comm_unique_id() if rank == 0 else my_id),
                                           dtype=torch.uint8).to(dev)
                    dist.broadcast(idt, src=0)
                    sv.comm_init(bytes(idt.cpu().numpy().tobytes()), rank, world)
                gather_path = f"library: tpc_mpc_solve_batch_compact_sharded_split, {a.split} split (ncclAllGather over RCCL)"
            except Exception as exc:   # noqa: BLE001
                probe_err = str(exc)
        if probe_err:
            print(f"[bench] rank {rank}: library RCCL path unavailable ({probe_err}); gathering with torch.distributed",
                  file=sys.stderr, flush=True)
            gather_path = f"torch.distributed all_gather (library RCCL path failed: {probe_err})"
        ok = torch.tensor([1 if gather_path.startswith("library") else 0], device=dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)   # all ranks take the same path
        if int(ok.item()) == 0 and gather_path.startswith("library"):
            gather_path = "torch.distributed all_gather (another rank's library RCCL path failed)"
    use_lib_gather = gather_path.startswith("library")
    # what produced the gathered outputs, as ONE top-level word: a scaling line made without the library's own
    # RCCL exchange must not look like a library number
    gather_word = "none" if world == 1 else ("library" if use_lib_gather else "fallback")
    if gather_word == "fallback" and not a.allow_fallback_gather:
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        raise SystemExit(f"[bench] rank {rank}: the library's RCCL gather is unavailable ({gather_path}); "
                         f"refusing to time a torch.distributed gather (pass --allow-fallback-gather to do so)")

    def step(k):
        i = k % slots
        with torch.cuda.stream(streams[i]):
            if world > 1 and use_lib_gather:
                solvers[i].solve_batch_compact_sharded(n_total, tv, ty, tp, out=(front_all[i], rear_all[i]), split=a.split)
            elif world > 1:
                if a.split != "block":
                    raise SystemExit("[bench] the torch.distributed fallback gather knows the block split only")
                lo = rank * n
                solvers[i].solve_batch_compact(tv, ty, tp, out=(front_all[i][lo:lo + n], rear_all[i][lo:lo + n]),
                                               want_flags=False)
                dist.all_gather_into_tensor(front_all[i], front_all[i][lo:lo + n].clone())
                dist.all_gather_into_tensor(rear_all[i], rear_all[i][lo:lo + n].clone())
            else:
                solvers[i].solve_batch_compact(tv, ty, tp, out=(fronts[i], rears[i]), want_flags=False)

    def barrier():
        if world > 1:
            dist.barrier()

    def kernel_times(used):
        """Mean (first, second) kernel duration in ms over the last solve of each used handle."""
        x1 = x2 = 0.0
        algo = 0
        for i in range(used):
            t1, t2, algo = solvers[i].last_kernel_times()
            x1 += t1 / used
            x2 += t2 / used
        return x1, x2, algo

    # RCCL communicator and all-gather channels are set up outside any step (a one-element-per-rank
    # collective, not workload); everything else is warmed by the W warmup steps
    if world > 1:
        prime = torch.zeros(world, dtype=tdt, device=dev)
        dist.all_gather_into_tensor(prime, prime[rank:rank + 1].clone())
    torch.cuda.synchronize()

    for k in range(a.warmup):
        step(k)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(a.steps):
        step(k)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    local_elapsed = elapsed
    # the same K-step region twice more (same bracket): box-to-box and run-to-run noise is +-4 %, so a 3 % kernel change is
    # only visible in the median; `value` stays the FIRST region, the one the contract describes
    region_s = [elapsed]
    for _ in range(2):
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for k in range(a.steps):
            step(k)
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        region_s.append(time.perf_counter() - t1)
    if world > 1:
        tr = torch.tensor(region_s, dtype=torch.float64, device=dev)
        dist.all_reduce(tr, op=dist.ReduceOp.MAX)
        region_s = [float(x) for x in tr.tolist()]
    # HIP events the library recorded on the launch streams inside the timed region
    k1, k2, algo_ran = kernel_times(min(slots, a.steps))
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        front, rear = front_all[0][my], rear_all[0][my]
        # the gather really delivered the other ranks' blocks: every rank checks the checksum of the
        # whole output against the all-reduced sum of the per-rank block checksums (exact: integers)
        chk = lambda t: t.view(torch.int64 if t.dtype == torch.float64 else torch.int32).bitwise_and(0xFFFFF).sum()
        local = torch.stack([chk(front), chk(rear)])
        dist.all_reduce(local, op=dist.ReduceOp.SUM)
        whole = torch.stack([chk(front_all[0]), chk(rear_all[0])])
        gather_ok = bool(torch.equal(local, whole))
        print(f"[bench] rank {rank}/{world}: instances {my.start}:{my.stop}:{my.step or 1} of {n_total}; "
              f"gather {'verified' if gather_ok else 'FAILED'} ({gather_path})", file=sys.stderr, flush=True)
    else:
        front, rear = fronts[0], rears[0]
        gather_ok = None
        chk = lambda t: t.view(torch.int64 if t.dtype == torch.float64 else torch.int32).bitwise_and(0xFFFFF).sum()

    # One record per rank, in the line: device, shard, iteration total (= the rank's share of the work; the projected-
    # gradient kernel's time follows it), kernel times, wall time, checksum of its block of the outputs -- so that the first
    # run on a multi-GPU node can be diagnosed from its one line.  Outside the timed region.
    _, _, it_r = solver.solve_batch_compact(tv, ty, tp, want_iters=True)
    torch.cuda.synchronize()
    rec = {"rank": rank, "device": local_rank, "device_name": torch.cuda.get_device_name(dev),
           "shard": {"first": my.start, "stride": my.step or 1, "count": n},
           "iterations_total": int(it_r.sum().item()), "cd_kernel_ms": k1, "pg_kernel_ms": k2,
           "ms_per_step_local": local_elapsed / a.steps * 1e3,
           "block_checksum": [int(chk(front).item()), int(chk(rear).item())], "gather_verified": gather_ok}
    recs = [rec]
    if world > 1:
        recs = [None] * world
        dist.all_gather_object(recs, rec)
    it_tot = [r["iterations_total"] for r in recs]
    imbalance = {"iterations_max_over_mean": max(it_tot) / (sum(it_tot) / len(it_tot)), "split": a.split,
                 "sorted_by_speed": bool(a.sorted_by_speed)}

    if rank == 0:
        # iteration statistics of this rank's shard (for the algorithmic-flop figure)
        s1 = s2 = 0.0
        for _ in range(3):   # one batch in flight: kernel durations without any overlap
            _, _, iters_t = solver.solve_batch_compact(tv, ty, tp, want_iters=True)
            y1, y2, _ = solver.last_kernel_times()
            s1 += y1 / 3
            s2 += y2 / 3
        torch.cuda.synchronize()
        mean_iters = float(iters_t.double().mean().item())
        lane_stats = None
        if algo_ran in (2, 3, 4):
            wi, rb = solver.last_lane_stats()
            pg_iters = float((iters_t.double() - 50.0).clamp(min=0).sum().item())
            lane_stats = {"wave_iterations": wi, "refill_blocks": rb}
            if algo_ran != 4:   # (a GROUP wavefront carries 64 / G instances: the figure has no such meaning there)
                lane_stats["lane_utilisation"] = pg_iters / (64.0 * wi) if wi else None
        esz = 8 if a.dtype == "f64" else 4
        total = world * n * a.steps
        value = total / elapsed
        # dominant kernel = the longer of the two launches of a step
        dom_ms = max(k1, k2)
        dom_name = {1: "wave_kernel", 2: "lane_pg_fused_kernel" if k2 >= k1 else "lane_cd_kernel",
                    3: ("ub_pg_asm_kernel" if (a.dtype == "f64" and H == 20) else "ub_pg_kernel") if k2 >= k1 else "ub_cd_kernel",   # (fp64 N = 20: the hand-written kernel, csrc/mpc_ub_asm.h)
                    4: "group_pg_kernel" if k2 >= k1 else "ub_cd_kernel"}[algo_ran]
        alg_bytes = 5 * esz * n                       # 3 in + 2 out scalars per solve
        # the PG kernel also reads what the CD kernel left per instance (not algorithmic traffic)
        hbm_gbs = alg_bytes / (dom_ms * 1e-3) / 1e9
        alg_flops = (46 * H - 16) * mean_iters * n
        tfl = alg_flops / (elapsed / a.steps) / 1e12                 # what a whole step delivers
        tfl_serial = alg_flops / ((s1 + s2) * 1e-3) / 1e12             # the two kernels on their own
        peak_tf = FP64_VECTOR_PEAK_TF if a.dtype == "f64" else FP32_VECTOR_PEAK_TF
        # HBM bytes per launch of the dominant kernel from the PMC passes (scripts/profile.sh ->
        # scripts/make_traffic.py).  rocprofv3 cannot run inside this process, so the figure is taken from
        # profiles/traffic.json -- but only if that file was made from THIS binary (sha256 of the .so).
        traffic, traffic_note = None, "no profiles/traffic.json entry for this kernel"
        tj = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tj):
            try:
                tdata = json.load(open(tj))
                entry = tdata.get(f"{dom_name}_{a.dtype}_H{H}_n{n}")
                if entry is not None and tdata.get("_library_sha256") == library_sha256():
                    traffic, traffic_note = entry, "PMC passes of this binary (profiles/traffic.json)"
                elif entry is not None:
                    traffic_note = ("profiles/traffic.json was measured on a different build of libtpc_mpc.so "
                                    f"({entry} B per launch there): not reported for this binary")
            except Exception as exc:   # noqa: BLE001
                traffic_note = f"profiles/traffic.json unreadable: {exc}"
        out = {
            "metric": "MPC QP solves/sec (horizon N=20, 2 inputs) at 1/2/4/8 MI355X; max|du| vs dlib",
            "value": value, "unit": "solves/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": elapsed / a.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "value_median_of": {"repeats": len(region_s), "value": total / sorted(region_s)[len(region_s) // 2],
                                "values": [total / t for t in region_s]},
            "vs_baseline": None, "dtype": a.dtype, "data": "synthetic", "gather": gather_word,
            "config": {"workload": f"batch {n} trajectories per GPU, N={H}, 2 inputs, compact "
                                   f"(mpcControllerTobi) form, cold start, eps 0.01, max_iter 10000",
                       "global_batch": world * n, "horizon": H, "algo": {1: "wave", 2: "lane", 3: "lane_fma", 4: "group"}[algo_ran],
                       "parallelism": f"batch-sharded x{world}", "batches_in_flight": slots,
                       "gather": gather_path, "gather_verified": gather_ok},
            "build": MpcSolver.build_info(),
            "ranks": recs, "imbalance": imbalance,
            "kernel_ms": {"first": k1, "second": k2, "dominant": dom_name},
            "kernel_ms_serial": {"first": s1, "second": s2},
            "mean_iterations": mean_iters, "lane_stats": lane_stats,
            "roofline": {"bound": "hbm", "achieved": hbm_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": hbm_gbs / HBM_PEAK_GBS, "traffic": traffic, "traffic_note": traffic_note,
                         "note": "40 B/solve algorithmic: this path is issue-bound, not HBM-bound; see alu"},
            "alu": {"bound": "fp64 vector issue" if a.dtype == "f64" else "fp32 vector issue",
                    "achieved": tfl, "peak": peak_tf, "unit": "TFLOP/s", "frac": tfl / peak_tf,
                    "serial_kernels_frac": tfl_serial / peak_tf,
                    "flops_per_solve": (46 * H - 16) * mean_iters},
        }
        if not a.no_cpu and world == 1:   # (the CPU baseline and the side legs below belong to the N = 1 line only)
            cb, ns, cf, cr = cpu_baseline(H, v, dy, dphi, a.cpu_seconds)
            gf, gr = front[:ns].cpu().numpy().astype(np.float64), rear[:ns].cpu().numpy().astype(np.float64)
            out["cpu_baseline"] = cb
            out["max_abs_du_vs_dlib"] = float(max(np.abs(gf - cf).max(), np.abs(gr - cr).max()))
            # relative to dlib's own output (north_star: within 1e-6 rel); where dlib returns exactly 0 the
            # absolute difference is taken against the bound's magnitude instead
            den_f = np.where(cf != 0, np.abs(cf), 22 * np.pi / 180)
            den_r = np.where(cr != 0, np.abs(cr), 22 * np.pi / 180)
            out["max_rel_du_vs_dlib"] = float(max((np.abs(gf - cf) / den_f).max(), (np.abs(gr - cr) / den_r).max()))
            out["dlib_sample"] = ns
            out["gpu_over_cpu"] = value / cb["value"]
        if world == 1 and a.dtype == "f64" and algo_ran == 3 and not a.no_bit_exact:
            # the bit-exact LANE family on the same K steps, and the whole batch of the two families side by side
            with MpcSolver(horizon=H, device=local_rank, dtype="f64", algo="lane") as sl:
                sl.set_profiling(True)
                sl.reserve(n)
                lf, lr = torch.empty_like(tv), torch.empty_like(tv)
                for _ in range(max(1, a.warmup)):
                    sl.solve_batch_compact(tv, ty, tp, out=(lf, lr), want_flags=False)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(a.steps):
                    sl.solve_batch_compact(tv, ty, tp, out=(lf, lr), want_flags=False)
                torch.cuda.synchronize()
                dl = time.perf_counter() - t1
                b1, b2, _ = sl.last_kernel_times()
                _, _, lit = sl.solve_batch_compact(tv, ty, tp, want_iters=True)
                torch.cuda.synchronize()
            be = {"algo": "lane", "value": n * a.steps / dl, "unit": "solves/s", "ms_per_step": dl / a.steps * 1e3,
                  "kernel_ms": {"first": b1, "second": b2},
                  "alu_frac": alg_flops / (dl / a.steps) / 1e12 / peak_tf}
            if not a.no_cpu:
                lfc, lrc = lf[:ns].cpu().numpy(), lr[:ns].cpu().numpy()
                be["max_abs_du_vs_dlib"] = float(max(np.abs(lfc - cf).max(), np.abs(lrc - cr).max()))
            out["bit_exact"] = be
            out["vs_bit_exact"] = {
                "instances": n,
                "iteration_counts_equal": float((iters_t == lit).double().mean().item()),
                "max_abs_du": float(torch.maximum((front - lf).abs().max(), (rear - lr).abs().max()).item())}
        if world == 1 and slots == 1 and not a.no_pipelined:
            # the same K steps with two batches in flight (two handles, two streams): the next batch's
            # coordinate-descent kernel and queue sort run while the previous batch's last long instances
            # finish.  Reported beside `value`, which stays the one-batch-at-a-time figure whose kernel
            # durations agree with the rocprofv3 summaries under profiles/.
            s2 = MpcSolver(horizon=H, device=local_rank, dtype=a.dtype, algo=a.algo)
            s2.reserve(n)
            pair = [(solver, streams[0], fronts[0], rears[0]),
                    (s2, torch.cuda.Stream(dev), torch.empty_like(tv), torch.empty_like(tv))]

            def pstep(k):
                sv, st, fo, ro = pair[k % 2]
                with torch.cuda.stream(st):
                    sv.solve_batch_compact(tv, ty, tp, out=(fo, ro), want_flags=False)
            for k in range(4):
                pstep(k)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for k in range(a.steps):
                pstep(k)
            torch.cuda.synchronize()
            dp = time.perf_counter() - t1
            same = bool(torch.equal(pair[1][2], fronts[0]) and torch.equal(pair[1][3], rears[0]))
            out["pipelined"] = {"value": n * a.steps / dp, "unit": "solves/s", "batches_in_flight": 2,
                                "ms_per_step": dp / a.steps * 1e3, "outputs_identical_to_serial": same,
                                "alu_frac": alg_flops / (dp / a.steps) / 1e12 / peak_tf}
            s2.close()
        if world == 1 and not a.no_fp32 and a.dtype == "f64":
            s32 = MpcSolver(horizon=H, device=local_rank, dtype="f32", algo=a.algo)
            s32.set_profiling(True)
            s32.reserve(n)
            v32, y32, p32 = tv.float(), ty.float(), tp.float()
            f32, r32 = torch.empty_like(v32), torch.empty_like(v32)
            for _ in range(max(1, a.warmup)):
                s32.solve_batch_compact(v32, y32, p32, out=(f32, r32), want_flags=False)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(a.steps):
                s32.solve_batch_compact(v32, y32, p32, out=(f32, r32), want_flags=False)
            torch.cuda.synchronize()
            d32 = time.perf_counter() - t1
            c1, c2, c_algo = s32.last_kernel_times()
            _, _, it32 = s32.solve_batch_compact(v32, y32, p32, want_iters=True)
            err = torch.maximum((f32.double() - front).abs(), (r32.double() - rear).abs())
            mi32 = float(it32.double().mean().item())
            a32 = {1: "wave", 2: "lane", 3: "lane_fma", 4: "group"}.get(c_algo, str(c_algo))
            dom32 = {1: "wave_kernel", 2: "lane_pg_fused_kernel", 3: "ub_pg_kernel", 4: "group_pg_kernel"}.get(c_algo, "?")
            t32, t32_note = None, "no profiles/traffic.json entry for this kernel"
            try:
                tdata = json.load(open(tj))
                e32 = tdata.get(f"{dom32}_f32_H{H}_n{n}")
                if e32 is not None and tdata.get("_library_sha256") == library_sha256():
                    t32, t32_note = e32, "PMC passes of this binary (profiles/traffic.json)"
                elif e32 is not None:
                    t32_note = f"profiles/traffic.json was measured on a different build ({e32} B per launch there)"
            except Exception as exc:   # noqa: BLE001
                t32_note = f"profiles/traffic.json unreadable: {exc}"
            gbs32 = 5 * 4 * n / (max(c1, c2) * 1e-3) / 1e9             # 3 in + 2 out floats per solve
            tf32 = (46 * H - 16) * mi32 * n / (d32 / a.steps) / 1e12
            out["fp32"] = {"value": n * a.steps / d32, "unit": "solves/s (1 GPU)",
                           "ms_per_step": d32 / a.steps * 1e3, "algo": a32,
                           "kernel_ms": {"first": c1, "second": c2, "dominant": dom32},
                           "mean_iterations": mi32,
                           "roofline": {"bound": "hbm", "achieved": gbs32, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                        "frac": gbs32 / HBM_PEAK_GBS, "traffic": t32, "traffic_note": t32_note,
                                        "note": "20 B/solve algorithmic (config 3 as written: fp32)"},
                           "alu": {"bound": "fp32 vector issue", "achieved": tf32, "peak": FP32_VECTOR_PEAK_TF, "unit": "TFLOP/s",
                                   "frac": tf32 / FP32_VECTOR_PEAK_TF, "flops_per_solve": (46 * H - 16) * mi32,
                                   "note": "the 157.3 TFLOP/s fp32 vector peak assumes packed FMAs; a scalar-fp32 stream tops out at half"},
                           "within": {str(t): float((err <= t).double().mean().item())
                                      for t in (1e-2, 1e-3, 1e-4, 1e-5, 1e-6, 1e-7, 1e-8, 1e-9)},
                           "note": "fp32 has no reference (dlib is fp64-only): errors are against this run's fp64 outputs",
                           "max_abs_du_vs_fp64": float(err.max().item())}
            s32.close()
        if world == 1 and not a.no_config2 and a.dtype == "f64" and n == 262144 and H == 20:
            # BASELINE config 2 beside the headline: 4 096 instances at N = 10 through the WAVE family (the same
            # workload `--batch 4096 --horizon 10 --algo wave` times on its own; profiles/r02_config2_*)
            c2 = [torch.from_numpy(x).to(dev) for x in compact_inputs(10, 4096)]
            with MpcSolver(horizon=10, device=local_rank, dtype="f64", algo="wave") as sw:
                for _ in range(3):
                    sw.solve_batch_compact(*c2, want_flags=False)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(50):
                    sw.solve_batch_compact(*c2, want_flags=False)
                torch.cuda.synchronize()
                d2 = time.perf_counter() - t1
            out["config2"] = {"workload": "batch 4096, N=10, fp64, WAVE family", "value": 4096 * 50 / d2,
                              "unit": "solves/s", "ms_per_step": d2 / 50 * 1e3}
        if world == 1 and not a.no_mid and a.dtype == "f64" and n == 262144 and H == 20:
            # A mid-size batch beside the headline: 16 384 instances of the same workload through AUTO -- too many for a
            # wavefront each, too few to fill the chip one lane each: the GROUP family's place (G lanes per instance)
            cm = [torch.from_numpy(x).to(dev) for x in compact_inputs(20, 16384)]
            with MpcSolver(horizon=20, device=local_rank, dtype="f64", algo="auto") as sg:
                sg.set_profiling(True)
                for _ in range(3):
                    sg.solve_batch_compact(*cm, want_flags=False)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(20):
                    sg.solve_batch_compact(*cm, want_flags=False)
                torch.cuda.synchronize()
                dm = time.perf_counter() - t1
                m1, m2, m_algo = sg.last_kernel_times()
                _, _, mit = sg.solve_batch_compact(*cm, want_iters=True)
                torch.cuda.synchronize()
            mflops = (46 * 20 - 16) * float(mit.double().sum().item())
            out["mid_batch"] = {"workload": "batch 16384, N=20, fp64, AUTO", "algo": {1: "wave", 2: "lane", 3: "lane_fma", 4: "group"}.get(m_algo, str(m_algo)),
                                "value": 16384 * 20 / dm, "unit": "solves/s", "ms_per_step": dm / 20 * 1e3,
                                "kernel_ms": {"first": m1, "second": m2},
                                "alu_frac": mflops / (dm / 20) / 1e12 / FP64_VECTOR_PEAK_TF}
            # the same batch through the bit-exact family (at this size: G lanes per instance, csrc/mpc_lanex.h)
            with MpcSolver(horizon=20, device=local_rank, dtype="f64", algo="lane") as sb:
                for _ in range(2):
                    sb.solve_batch_compact(*cm, want_flags=False)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(10):
                    sb.solve_batch_compact(*cm, want_flags=False)
                torch.cuda.synchronize()
                out["mid_batch"]["bit_exact_ms_per_step"] = (time.perf_counter() - t1) / 10 * 1e3
        if world == 1 and not a.no_config5 and a.dtype == "f64" and n == 262144 and H == 20:
            # BASELINE config 5 beside the headline: 65 536 instances split evenly over N in {5, 10, 20, 40}, interleaved,
            # ONE tpc_mpc_solve_batch_compact_mixed call per step (binned on the device; under AUTO the bins run one after
            # another, each with the whole chip: tpc_mpc_mixed.hip)
            Hs, per = (5, 10, 20, 40), 16384
            parts = [compact_inputs(Hh, per) for Hh in Hs]
            mv, my, mp = (np.concatenate([pp[c] for pp in parts]) for c in range(3))
            hz = np.repeat(np.array(Hs, dtype=np.int32), per)
            perm = np.random.default_rng(3).permutation(len(hz))
            c5 = {"workload": "batch 65536 mixed over N in {5,10,20,40}, one call", "unit": "ms per batch",
                  "note": "f64: AUTO as shipped -- the ~10 % of the N=40 instances that end on the 10 000-iteration cap are "
                          "solved once more bit-exactly (10 000 iterations of dlib's own arithmetic, eight lanes per instance: "
                          "csrc/mpc_lanex.h); "
                          "f64_fast_capped: TPC_MPC_PARAM_FAST_CAPPED, the tolerance families' answer kept (<= 1.5e-12 from "
                          "dlib on this workload)"}
            from trajectory_controller_amd import capi
            # the reference's answers for the first 1 024 instances of each horizon's stream: the committed real-dlib fixtures
            # (outputs) and, for the iteration counts the fixtures only bound from below, the checker run on the host
            fix = {}
            if not a.no_cpu:
                from oracle import bindings as ob5
                for Hh, pp in zip(Hs, parts):
                    g = np.load(os.path.join(ROOT, "tests", "golden", f"compact_H{Hh}.npz"))
                    m = len(g["front"])
                    of5, or5, oi5 = ob5.Oracle().solve_compact(Hh, pp[0][:m], pp[1][:m], pp[2][:m], nthreads=os.cpu_count() or 1)
                    assert np.array_equal(of5, g["front"]) and np.array_equal(or5, g["rear"])   # (the checker IS dlib, bit for bit)
                    fix[Hh] = (g["front"], g["rear"], oi5)
            for dt_name, dt_t, opt in (("f64", torch.float64, 0), ("f64_fast_capped", torch.float64, capi.PARAM_FAST_CAPPED),
                                       ("f32", torch.float32, 0)):
                xv, xy, xp = (torch.from_numpy(z[perm]).to(dev, dtype=dt_t) for z in (mv, my, mp))
                with MpcSolver(horizon=20, device=local_rank, dtype=dt_name[:3], options=opt) as sm:
                    sm.solve_batch_compact_mixed(hz[perm], xv, xy, xp)
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    for _ in range(5):
                        sm.solve_batch_compact_mixed(hz[perm], xv, xy, xp)
                    torch.cuda.synchronize()
                    c5[dt_name] = (time.perf_counter() - t1) / 5 * 1e3
                    # parity of this very batch: the first 1 024 instances of every horizon's stream are the real-dlib
                    # fixtures (tests/golden/compact_H*.npz) -- relative error histogram and iteration-count agreement per
                    # horizon
                    if fix:
                        f5, r5, i5 = sm.solve_batch_compact_mixed(hz[perm], xv, xy, xp, want_iters=True)
                        torch.cuda.synchronize()
                        inv = np.empty_like(perm)
                        inv[perm] = np.arange(len(perm))
                        f5, r5, i5 = (z.cpu().numpy()[inv] for z in (f5, r5, i5))
                        par = {}
                        for bi, Hh in enumerate(Hs):
                            gf5, gr5, gi5 = fix[Hh]
                            m = len(gf5)
                            sl5 = slice(bi * per, bi * per + m)
                            bound = 22 * np.pi / 180
                            rel = np.maximum(np.abs(f5[sl5].astype(np.float64) - gf5) / np.where(gf5 != 0, np.abs(gf5), bound),
                                             np.abs(r5[sl5].astype(np.float64) - gr5) / np.where(gr5 != 0, np.abs(gr5), bound))
                            par[f"N={Hh}"] = {"instances": int(m),
                                              "rel_err_within": {str(t): float((rel <= t).mean()) for t in
                                                                 (1e-2, 1e-3, 1e-4, 1e-5, 1e-6, 1e-7, 1e-8, 1e-9)},
                                              "max_rel_err": float(rel.max()),
                                              "iteration_counts_equal": float((i5[sl5] == gi5).mean())}
                        c5.setdefault("vs_dlib_fixtures", {})[dt_name] = par
            c5["solves_per_s_f64"] = len(hz) / (c5["f64"] * 1e-3)
            c5["solves_per_s_f64_fast_capped"] = len(hz) / (c5["f64_fast_capped"] * 1e-3)
            out["config5"] = c5
        if world == 1 and not a.no_config1 and not a.no_cpu and a.dtype == "f64" and n == 262144 and H == 20:
            # BASELINE config 1 beside the headline: ONE trajectory, N=10 -- the call the reference's cycle() makes
            # (mpcControllerTobi -> tpc_mpc_solve_one), with a new speed in every call: on the calling thread (the host
            # path, TPC_MPC_OPT_HOST_SOLVE_ONE), through the resident wavefront, and real dlib on one host core beside
            # them (the checker library, after everything timed above).  Medians over 1 000 calls, Python's ctypes call
            # included in all three (~1-2 us; examples/solve_one_latency.c measures the first two without it).
            from oracle import bindings as ob
            from trajectory_controller_amd import capi as _capi

            def med_us(call, reps=1000):
                ts = []
                for i in range(reps + 50):
                    t1 = time.perf_counter()
                    call(1.0 + 2e-3 * i)
                    ts.append(time.perf_counter() - t1)
                return float(np.median(np.array(ts[50:])) * 1e6)
            c1 = {"workload": "one trajectory, N=10, fp64: tpc_mpc_solve_one (what mpcControllerTobi calls)", "unit": "us per solve (median)",
                  "note": "speeds 1.1 .. 3.1 m/s, target (0.1, 0.05): a single short-horizon solve is latency, not throughput -- a host core "
                          "is as fast as the round trip to the GPU here (SURVEY.md section 7 said so); the batch entries are the product"}
            with MpcSolver(horizon=10, device=local_rank) as so:
                c1["resident_wavefront"] = med_us(lambda vv: so.mpc_controller_tobi(vv, 0.1, 0.05))
                gres = so.mpc_controller_tobi(2.0, 0.1, 0.05)
                so.set_option(_capi.OPT_HOST_SOLVE_ONE, 10)
                c1["calling_thread"] = med_us(lambda vv: so.mpc_controller_tobi(vv, 0.1, 0.05))
                hres = so.mpc_controller_tobi(2.0, 0.1, 0.05)
            if os.path.exists(ob.REF_SO):
                # (dlib: the same 1 000 speeds as one single-threaded batch call -- model build, mpc constructor, set_target
                # and operator() per instance as in cycle() -- so that no per-call binding overhead lands on the reference)
                ref1 = ob.DlibRef(ob.REF_SO)
                vs = 1.0 + 2e-3 * np.arange(50, 1050)
                ref1.solve_compact(10, vs[:64], np.full(64, 0.1), np.full(64, 0.05), nthreads=1)
                t1 = time.perf_counter()
                ref1.solve_compact(10, vs, np.full(vs.size, 0.1), np.full(vs.size, 0.05), nthreads=1)
                c1["dlib_one_core"] = (time.perf_counter() - t1) / vs.size * 1e6
                df, dr = ref1.solve_compact(10, np.array([2.0]), np.array([0.1]), np.array([0.05]), nthreads=1)
                c1["max_abs_du_vs_dlib"] = float(max(abs(gres[0] - df[0]), abs(gres[1] - dr[0]), abs(hres[0] - df[0]), abs(hres[1] - dr[0])))
            out["config1"] = c1
        if world == 1 and not a.no_config4 and a.dtype == "f64" and n == 262144 and H == 20:
            # BASELINE config 4's workload beside the headline: its WHOLE batch -- 2 097 152 trajectories, N=20, fp32 as
            # written -- on this ONE GPU.  Not a scaling number (that is `--gpus 8`, eight of these blocks side by side plus
            # the all-gather of 2 x 8 MiB per rank): what the eight ranks' arithmetic costs when one card does all of it.
            n4 = 8 * n
            c4 = [torch.from_numpy(x).to(dev, dtype=torch.float32) for x in compact_inputs(20, n4)]
            with MpcSolver(horizon=20, device=local_rank, dtype="f32", algo=a.algo) as s4:
                s4.set_profiling(True)
                s4.reserve(n4)
                f4, r4 = torch.empty_like(c4[0]), torch.empty_like(c4[0])
                for _ in range(2):
                    s4.solve_batch_compact(*c4, out=(f4, r4), want_flags=False)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(5):
                    s4.solve_batch_compact(*c4, out=(f4, r4), want_flags=False)
                torch.cuda.synchronize()
                d4 = (time.perf_counter() - t1) / 5
                q1, q2, q_algo = s4.last_kernel_times()
            out["config4_one_gpu"] = {"workload": "batch 2097152 (config 4's whole batch), N=20, fp32, ONE GPU", "value": n4 / d4,
                                      "unit": "solves/s", "ms_per_step": d4 * 1e3,
                                      "algo": {1: "wave", 2: "lane", 3: "lane_fma", 4: "group"}.get(q_algo, str(q_algo)),
                                      "kernel_ms": {"first": q1, "second": q2},
                                      # (the fp32 leg above solves this first block by itself, under AUTO in another kernel
                                      # family at that size: the same arithmetic in another association)
                                      "max_abs_du_first_block_vs_fp32_leg": (float(torch.maximum((f4[:n] - f32).abs().max(), (r4[:n] - r32).abs().max()).item())
                                                                             if "fp32" in out else None)}
            del c4, f4, r4
        print(json.dumps(out), flush=True)
    for sv in solvers:
        sv.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
        if not gather_ok:
            raise SystemExit(f"[bench] rank {rank}: the gathered outputs do not contain every rank's block")


if __name__ == "__main__":
    main()
