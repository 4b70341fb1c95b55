"""msorb_host::KeyFrameStore (ms-slam_amd/host/ORBmatcher_device.h) under concurrent Ensure / Forget / Reset from four threads:
tests/kfstore_concurrency_main.cc against the stand-ins of tests/slam_stub.  ADVICE round 2: the store dropped its lock between
lookup and insert and removed ids other threads were about to search with."""
import json
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def test_concurrent_ensure_forget_reset(tmp_path):
    exe = tmp_path / "kfstore_concurrency"
    subprocess.check_call(["g++", "-std=c++17", "-O2", f"-I{ROOT}/tests/slam_stub", f"-I{ROOT}/tests/cv_stub", f"-I{ROOT}/ms-slam_amd/host",
                           f"-I{ROOT}/include", f"{ROOT}/tests/kfstore_concurrency_main.cc", f"-L{ROOT}/ms-slam_amd", "-lmsorb",
                           f"-Wl,-rpath,{ROOT}/ms-slam_amd", "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib", "-lpthread", "-o", str(exe)])
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["mismatches"] == 0 and out["exceptions"] == 0 and out["recycled_id_readded"] == 1
    assert out["device_entries"] == out["resident"] + 1
