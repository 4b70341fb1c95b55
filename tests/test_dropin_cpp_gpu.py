"""Compiles the drop-in C++ class ORB_SLAM3::ORBextractor (ms-slam_amd/host, same declaration as the
reference's include/ORBextractor.h) against the test-only cv stub, links libmsorb.so, runs it like
Frame::ExtractORB (Frame.cc:418-425) and compares with the oracle."""
import os
import struct
import subprocess

import numpy as np
import pytest

from msorb import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def test_cpp_dropin_matches_oracle(tmp_path, oracle, msorb_mod):
    exe = tmp_path / "dropin_extractor"
    subprocess.check_call(["g++", "-std=c++17", "-O2", f"-I{ROOT}/tests/cv_stub", f"-I{ROOT}/ms-slam_amd/host",
                           f"-I{ROOT}/include", f"{ROOT}/tests/dropin_extractor_main.cc",
                           f"{ROOT}/ms-slam_amd/host/ORBextractor.cc", f"-L{ROOT}/ms-slam_amd", "-lmsorb",
                           f"-Wl,-rpath,{ROOT}/ms-slam_amd", "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib", "-o", str(exe)])
    cfg = synth.EUROC
    img = synth.image(77, cfg["rows"], cfg["cols"])
    raw, out = tmp_path / "in.raw", tmp_path / "out.bin"
    img.tofile(raw)
    subprocess.check_call([str(exe), str(cfg["rows"]), str(cfg["cols"]), str(raw), str(out), "1000"])
    blob = out.read_bytes()
    mono, n = struct.unpack_from("<ii", blob, 0)
    kps = np.frombuffer(blob, oracle.KP_DTYPE, n, 8)
    desc = np.frombuffer(blob, np.uint8, n * 32, 8 + 28 * n).reshape(n, 32)
    l7r, l7c, lv = struct.unpack_from("<iii", blob, 8 + 60 * n)
    rmono, rkps, rdesc = oracle.OracleExtractor(1000, 1.2, 8, 20, 7)(img)
    assert (mono, n) == (rmono, len(rkps)) and (l7r, l7c, lv) == (134, 210, 8)
    assert np.array_equal(kps.view(np.uint8), rkps.view(np.uint8)) and np.array_equal(desc, rdesc)


def test_cpp_matcher_adapter_matches_oracle(tmp_path, oracle, msorb_mod):
    """ms-slam_amd/host/ORBmatcher_device.h (the body a maintainer puts into ORBmatcher::SearchByProjection) compiled
    against stand-in Frame / MapPoint types, vs the oracle's restatement of ORBmatcher.cc:43-142."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import matcher_cases as mc
    exe = tmp_path / "dropin_matcher"
    subprocess.check_call(["g++", "-std=c++17", "-O2", f"-I{ROOT}/tests/cv_stub", f"-I{ROOT}/ms-slam_amd/host",
                           f"-I{ROOT}/include", f"{ROOT}/tests/dropin_matcher_main.cc", f"-L{ROOT}/ms-slam_amd", "-lmsorb",
                           f"-Wl,-rpath,{ROOT}/ms-slam_amd", "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib", "-o", str(exe)])
    cfg = synth.KITTI
    ex = msorb_mod.ORBextractor(cfg["nfeatures"], cfg["scale"], cfg["nlevels"], cfg["ini_th"], cfg["min_th"])
    try:
        _, kps, desc = ex(synth.image(5, cfg["rows"], cfg["cols"]))
        scale = ex.GetScaleFactors()
    finally:
        ex.close()
    rng = np.random.Generator(np.random.PCG64(9))
    N, M = len(kps), 3000
    ur = np.where(rng.random(N) < 0.6, kps["x"] - rng.uniform(1, 40, N), -1).astype(np.float32)
    mp = mc.map_point_table(rng, kps, desc, ur, scale, M, 0.2, 0.1)
    init = np.where(rng.random(N) < 0.25, rng.integers(0, M, N), -1).astype(np.int32)
    bounds = (0.0, float(cfg["cols"]), 0.0, float(cfg["rows"]))
    th, th_far, nnratio, far = 3.0, 60.0, 0.8, 1
    blob = tmp_path / "in.bin"
    with open(blob, "wb") as f:
        f.write(struct.pack("<iiii", N, len(scale), M, far))
        f.write(struct.pack("<7f", *bounds, th, th_far, nnratio))
        for a in (kps, desc, ur, np.asarray(scale, np.float32), mp["track_in_view"], mp["bad"], mp["sparsified"],
                  mp["proj_x"], mp["proj_y"], mp["proj_xr"], mp["track_depth"], mp["view_cos"], mp["level"], mp["obs"],
                  mp["desc"], init):
            f.write(np.ascontiguousarray(a).tobytes())
    out = tmp_path / "out.bin"
    subprocess.check_call([str(exe), str(blob), str(out)])
    res = np.frombuffer(out.read_bytes(), np.int32)
    nmatches, got = int(res[0]), res[1:]
    # oracle: the points the C++ side left out of the local map (held by the frame, odd index) are not visited
    held = np.zeros(M, bool)
    held[init[init >= 0]] = True
    mp_o = dict(mp)
    mp_o["track_in_view"] = np.where(held & (np.arange(M) % 2 == 1), 0, mp["track_in_view"]).astype(np.uint8)
    rf = oracle.OracleFrame(kps, desc, ur, bounds, scale)
    want = init.copy()
    rn = rf.SearchByProjection_mps(mp_o, want, th, bFarPoints=bool(far), thFarPoints=th_far, nnratio=nnratio)
    assert rn > 200
    assert nmatches == rn and np.array_equal(got, want)
