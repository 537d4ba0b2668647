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
    assert lines[-1] == {"idle_state": True, "priority": 100}
