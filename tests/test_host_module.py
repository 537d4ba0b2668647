"""The C++ module shim (trajectory_controller_amd/host): it compiles against the LMS stand-ins,
links the C ABI, refuses to initialise without a GPU (CPU test), and -- on the GPU -- cycle() writes
the steering angles dlib would have produced for the same (v, y_soll, phi_soll)."""
import json
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "trajectory_controller_amd", "host")
LIB = os.path.join(ROOT, "trajectory_controller_amd", "lib")


def _build_harness(tmp_path):
    subprocess.check_call(["make", "-C", HOST], stdout=subprocess.DEVNULL)
    exe = str(tmp_path / "module_harness")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I" + os.path.join(HOST, "lms_compat"), "-I" + HOST,
                           "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "host", "module_harness.cpp"),
                           "-o", exe, "-L" + LIB, "-ltrajectory_point_controller", "-ltpc_mpc",
                           "-Wl,-rpath," + LIB])
    return exe


def test_module_builds_and_fails_loudly_without_gpu(tmp_path):
    import torch
    exe = _build_harness(tmp_path)
    out = subprocess.run(["nm", "-D", "--defined-only", os.path.join(LIB, "libtrajectory_point_controller.so")],
                         capture_output=True, text=True).stdout
    assert " T getInstance" in out                       # LMS_MODULE_INTERFACE export (src/interface.cpp:3)
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: covered by the gpu test")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 3 and '"initialize": false' in r.stdout
    assert "tpc_mpc_create failed" in r.stderr            # no silent CPU fallback


@pytest.mark.gpu
@pytest.mark.parametrize("H", [4, 10])
def test_cycle_matches_reference_solver(tmp_path, oracle, H):
    exe = _build_harness(tmp_path)
    r = subprocess.run([exe, str(H)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    scen = [l for l in lines if "scenario" in l]
    assert len(scen) == 4 and all(s["ok"] for s in scen)
    v = np.array([s["v"] for s in scen])
    f, rr, _ = oracle.solve_compact(H, v, [s["y_soll"] for s in scen], [s["phi_soll"] for s in scen])
    for i, s in enumerate(scen):
        if s["targetSpeed"] < 0.5:
            assert s["steering_front"] == 0 and s["steering_rear"] == 0      # crossing rule (follower.cpp:277-283)
        else:
            assert abs(s["steering_front"] - f[i]) <= 1e-9 and abs(s["steering_rear"] - rr[i]) <= 1e-9
        assert s["driving"] == 1
    assert lines[-3] == {"indicator_left": True, "indicator_right": False, "reset_left": False, "reset_right": False}
    assert lines[-2] == {"idle_state": True, "priority": 100, "idle_speed": 0, "removed_after": True}
    assert lines[-1] == {"other_backend_refused": True}


def _traj_point_np(px, py, dx, dy, vel, count, want):
    """float32 restatement of the module shim's getTrajectoryPoint (host/trajectory_point_controller.cpp,
    reference src/trajectory_point_follower.cpp:392-443) for one polyline -- test infrastructure."""
    f = np.float32
    ox, oy, odx, ody, ovel = f(want), f(0), f(1), f(0), f(0)
    if count > 0:
        walked, found = f(0), False
        for i in range(1, count):
            ex, ey = f(px[i - 1] - px[i]), f(py[i - 1] - py[i])
            ln = f(np.sqrt(f(f(ex * ex) + f(ey * ey))))
            walked = f(walked + ln)
            if walked > want:
                back = f(walked - want)
                nx, ny = (f(ex / ln), f(ey / ln)) if ln > 0 else (f(0), f(0))
                ox, oy = f(px[i] + f(nx * back)), f(py[i] + f(ny * back))
                odx, ody, ovel, found = dx[i], dy[i], vel[i], True
                break
        if not found:
            j = count - 1
            ox, oy, odx, ody, ovel = px[j], py[j], dx[j], dy[j], vel[j]
    return ox, oy, odx, ody, ovel, f(np.sqrt(f(f(ox * ox) + f(oy * oy))))


@pytest.mark.gpu
def test_follow_batch_matches_module_and_oracle(tmp_path, oracle):
    """tpc_mpc_follow_batch (batched getTrajectoryPoint + target extraction + solve + crossing rule)
    against (1) the C++ module ticking the same trajectories one by one and (2) a float32 numpy
    restatement + the oracle on a large random batch with ragged point counts."""
    import torch
    from trajectory_controller_amd import MpcSolver
    H = 10
    exe = _build_harness(tmp_path)
    r = subprocess.run([exe, str(H)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    pts = [l for l in lines if "points" in l]
    scen = [l for l in lines if "scenario" in l]
    P = len(pts[0]["px"])
    dev = "cuda:0"
    t32 = lambda key: torch.tensor([p[key] for p in pts], dtype=torch.float32, device=dev).T.contiguous()
    cnt = torch.full((len(pts),), P, dtype=torch.int32, device=dev)
    carv = torch.tensor([p["car_velocity"] for p in pts], dtype=torch.float32, device=dev)
    look = torch.tensor([p["look_ahead"] for p in pts], dtype=torch.float32, device=dev)
    with MpcSolver(horizon=H) as s:
        f, rr, ts, td = s.follow_batch(t32("px"), t32("py"), t32("dx"), t32("dy"), t32("vel"), cnt, carv, look)
    for i, sc in enumerate(scen):   # the module, one cycle() at a time
        assert abs(float(f[i]) - sc["steering_front"]) <= 1e-9 and abs(float(rr[i]) - sc["steering_rear"]) <= 1e-9
        assert float(ts[i]) == np.float32(sc["targetSpeed"]) and float(td[i]) == np.float32(pts[i]["target_distance"])

    # large random batch, ragged counts (0, 1 and short polylines included), with a velocity lookup table
    rng = np.random.default_rng(11)
    n, P = 20000, 24
    seg = rng.uniform(0.02, 0.25, size=(P, n)).astype(np.float32)
    ang = np.cumsum(rng.uniform(-0.15, 0.15, size=(P, n)), axis=0).astype(np.float32)
    px = np.cumsum(seg * np.cos(ang), axis=0, dtype=np.float32)
    py = (np.cumsum(seg * np.sin(ang), axis=0, dtype=np.float32) + rng.uniform(-0.2, 0.2, size=n).astype(np.float32))
    dx, dy = np.cos(ang).astype(np.float32), np.sin(ang).astype(np.float32)
    vel = rng.uniform(0.0, 2.0, size=(P, n)).astype(np.float32)
    count = rng.integers(0, P + 1, size=n).astype(np.int32)
    count[:3] = (0, 1, 2)
    carv = rng.uniform(-0.2, 4.0, size=n).astype(np.float32)
    look = rng.uniform(0.2, 2.5, size=n).astype(np.float32)
    lut_x = np.array([0.0, 1.0, 2.5, 4.0], dtype=np.float32)
    lut_y = np.array([0.8, 1.0, 2.0, 2.4], dtype=np.float32)
    g = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    with MpcSolver(horizon=H) as s:
        f, rr, ts, td = s.follow_batch(g(px), g(py), g(dx), g(dy), g(vel), g(count), g(carv), g(look),
                                       lookup=(g(lut_x), g(lut_y)))
    f, rr, ts, td = (a.cpu().numpy() for a in (f, rr, ts, td))
    ev, ey, ephi, ets, etd = (np.zeros(n) for _ in range(5))
    for k in range(n):
        ox, oy, odx, ody, ovel, dist = _traj_point_np(px[:, k], py[:, k], dx[:, k], dy[:, k], vel[:, k], int(count[k]), look[k])
        vv = np.float64(carv[k])
        if abs(vv) < 0.1:
            vv = 0.1
        vv = np.float32(vv)
        # piecewise-linear table in float32, as the shim's LookupTable::linearSearch
        if vv <= lut_x[0]:
            lv = lut_y[0]
        elif vv > lut_x[-1]:
            lv = lut_y[-1]
        else:
            j = int(np.searchsorted(lut_x, vv, side="left"))
            t = np.float32((vv - lut_x[j - 1]) / np.float32(lut_x[j] - lut_x[j - 1]))
            lv = np.float32(lut_y[j - 1] + np.float32(t * np.float32(lut_y[j] - lut_y[j - 1])))
        ev[k], ey[k], ephi[k] = np.float64(lv), np.float64(oy), np.arctan2(np.float64(ody), np.float64(odx))
        ets[k], etd[k] = ovel, dist
    assert np.array_equal(ts, ets.astype(np.float32)) and np.array_equal(td, etd.astype(np.float32))
    of, orr, _ = oracle.solve_compact(H, ev, ey, ephi, nthreads=8)
    crossing = ets < 0.5
    of[crossing], orr[crossing] = 0.0, 0.0
    # atan2 on the device and in glibc may differ in the last bit, so steering is compared to 1e-9
    assert np.abs(f - of).max() <= 1e-9 and np.abs(rr - orr).max() <= 1e-9
    assert crossing.sum() > 100 and (~crossing).sum() > 100


def _build_example(tmp_path):
    exe = str(tmp_path / "batch_compact")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "batch_compact.c"), "-o", exe, "-L" + LIB, "-ltpc_mpc",
                           "-Wl,-rpath," + LIB])
    return exe


def test_c_example_builds_and_refuses_without_gpu(tmp_path):
    import torch
    exe = _build_example(tmp_path)
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: covered by the gpu test")
    r = subprocess.run([exe, "10", "8"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 3 and "tpc_mpc_create" in r.stderr


@pytest.mark.gpu
def test_c_example_matches_oracle(tmp_path, oracle):
    """A plain-C host, host-memory batch: the printed steering pairs are dlib's."""
    exe = _build_example(tmp_path)
    r = subprocess.run([exe, "10", "300"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    rows = [l.split() for l in r.stdout.splitlines() if "->" in l]
    assert len(rows) == 5
    val = lambda tok: float(tok.split("=")[1])
    v, dy, dphi = ([val(row[c]) for row in rows] for c in (1, 2, 3))
    f, rr, it = oracle.solve_compact(10, v, dy, dphi)
    for i, row in enumerate(rows):   # 300 instances: AUTO takes the WAVE kernel
        assert abs(val(row[5]) - f[i]) <= 1e-9 and abs(val(row[6]) - rr[i]) <= 1e-9 and int(val(row[7])) == it[i]
    assert "flags=0x0" in r.stdout
