"""Long-run evidence ("missing" 5 / "next" 4): tests/soak_main.cc — 20 000 tracking frames through the drop-in
classes on three threads (Tracking / LocalMapping / a sparsifier) with KeyFrame insertion, culling WITHOUT source edits (the store's
std::weak_ptr notices the dead object), sparsification flips and a budget phase; results held to single-threaded baselines that are
the oracle's; free device memory and ResidentKeyFrames() flat.  The harness decides; this file builds it, runs it and repeats its
verdict with the numbers."""
import json
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def build(tmp_path):
    exe = str(tmp_path / "soak")
    subprocess.check_call(["g++", "-O2", "-std=c++17", f"-I{ROOT}/tests/slam_stub", f"-I{ROOT}/tests/cv_stub", f"-I{ROOT}/ms-slam_amd/host",
                           f"-I{ROOT}/include", f"{ROOT}/tests/soak_main.cc", f"{ROOT}/ms-slam_amd/host/ORBextractor.cc",
                           f"{ROOT}/ms-slam_amd/host/ORBmatcher.cc", f"-L{ROOT}/ms-slam_amd", "-lmsorb", f"-Wl,-rpath,{ROOT}/ms-slam_amd",
                           f"-L{ROOT}/oracle", "-lorb_oracle", f"-Wl,-rpath,{ROOT}/oracle", "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib",
                           "-lpthread", "-o", exe])
    return exe


def test_soak_20000_tracking_frames_three_threads(tmp_path, oracle):
    exe = build(tmp_path)
    frames = int(os.environ.get("MSORB_SOAK_FRAMES", "20000"))
    p = subprocess.run([exe, str(frames), "10"], capture_output=True, text=True, timeout=1500)
    print(p.stdout[-3000:], p.stderr[-3000:])
    line = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert line, p.stderr[-2000:]
    r = json.loads(line[-1])
    assert r["probe_baseline_vs_oracle_mismatches"] == 0, "the single-threaded baselines are not the oracle's results"
    assert r["mismatches"] == 0 and r["exceptions"] == 0
    assert r["frames"] == frames and r["probes"] >= frames // 200 - 1 and r["local_mapping_calls"] > frames // 25 and r["sparsified"] > 0
    assert r["culled"] > r["culled_with_hook"] > 0 and r["expired_by_weak_ptr"] > 0, "no KeyFrame left the store through weak_ptr expiry"
    assert r["evicted_by_budget"] > 0 and r["resident_over_budget"] == 0
    assert r["resident_over_live"] == 0 and r["max_resident"] <= r["max_live_in_map"] + 10
    assert r["resident_after_map_dropped"] <= 2 and r["resident_end"] == 0
    assert r["free_memory_drift_bytes"] < 8 << 20, f"free device memory went down by {r['free_memory_drift_bytes']} bytes between the 2nd and the 4th quarter"
    assert p.returncode == 0 and r["ok"] is True
