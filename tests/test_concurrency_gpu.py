"""Concurrency parity (SURVEY.md §8b "Threading"): tests/concurrency_main.cc repeats the reference's threading shape — two
fresh threads per frame on two drop-in ORBextractor objects (Frame.cc:122-125) while three long-lived threads run
msorb_search_by_projection_mps / msorb_fuse_search / msorb_search_by_bow (Tracking.cc:2835, LocalMapping.cc:787,
LoopClosing.cc:594) — and compares every concurrent result with its single-threaded baseline; this driver checks the
baselines themselves against the oracle."""
import os
import struct
import subprocess
import sys

import numpy as np
import pytest

from msorb import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
pytestmark = pytest.mark.gpu


def test_two_eye_threads_and_three_matcher_threads(tmp_path, oracle, msorb_mod):
    import matcher_cases as mc
    import bow_match_cases as bmc
    exe = tmp_path / "concurrency"
    subprocess.check_call(["g++", "-std=c++17", "-O2", f"-I{ROOT}/tests/cv_stub", f"-I{ROOT}/ms-slam_amd/host", f"-I{ROOT}/include",
                           f"{ROOT}/tests/concurrency_main.cc", f"{ROOT}/ms-slam_amd/host/ORBextractor.cc", f"-L{ROOT}/ms-slam_amd",
                           "-lmsorb", f"-Wl,-rpath,{ROOT}/ms-slam_amd", "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib", "-lpthread",
                           "-o", str(exe)])
    cfg = synth.KITTI
    rows, cols = cfg["rows"], cfg["cols"]
    L, R = synth.stereo_pair(91, rows, cols)
    rng = np.random.Generator(np.random.PCG64(17))
    orc = oracle.OracleExtractor(2000, 1.2, 8, 20, 7)
    monoL, kl, dl = orc(L)
    monoR, kr, dr = orc(R)
    scale = orc.tables()["scale"].astype(np.float32)
    inv_sigma2 = (np.float32(1.0) / (scale * scale)).astype(np.float32)
    N = len(kl)
    ur = np.where(rng.random(N) < 0.6, kl["x"] - rng.uniform(1, 40, N), -1).astype(np.float32)
    M, Mf = 4096, 3000
    mp = mc.map_point_table(rng, kl, dl, ur, scale, M)
    frame_mp0 = np.where(rng.random(N) < 0.2, rng.integers(0, M, N), -1).astype(np.int32)
    src = rng.integers(0, N, Mf)
    fu = (kl["x"][src] + rng.normal(0, 1.0, Mf)).astype(np.float32)
    fv = (kl["y"][src] + rng.normal(0, 1.0, Mf)).astype(np.float32)
    fur = np.where(ur[src] >= 0, ur[src] + rng.normal(0, 1.0, Mf), fu - 20).astype(np.float32)
    flevel = np.clip(kl["octave"][src] + rng.integers(0, 2, Mf), 0, 7).astype(np.int32)
    fradius = (np.float32(3.0) * scale[flevel]).astype(np.float32)
    fvalid = (rng.random(Mf) < 0.9).astype(np.uint8)
    fdesc = mc.flip_bits(rng, dl[src], 40)
    bp = bmc.make_pair(5, n1=1500, n2=1800, n_nodes=80)
    with open(tmp_path / "in.bin", "wb") as f:
        f.write(struct.pack("<8i", rows, cols, N, 8, M, Mf, len(bp["desc1"]), len(bp["desc2"])))
        for a in (L, R, kl, dl, ur, scale, inv_sigma2, mp["track_in_view"], mp["bad"], mp["sparsified"], mp["proj_x"], mp["proj_y"],
                  mp["proj_xr"], mp["track_depth"], mp["level"], mp["view_cos"], mp["desc"], mp["obs"], frame_mp0, fvalid, fu, fv, fur,
                  flevel, fradius, fdesc, bp["desc1"], bp["desc2"], bp["valid1"], bp["avail2"]):
            f.write(np.ascontiguousarray(a).tobytes())
        for fvx in (bp["fv1"], bp["fv2"]):
            f.write(struct.pack("<i", len(fvx[0])))
            for a in fvx:
                f.write(np.ascontiguousarray(a, np.int32).tobytes())
        f.write(bp["angle1"].tobytes()); f.write(bp["angle2"].tobytes())
    iters = 100
    p = subprocess.run([str(exe), str(tmp_path / "in.bin"), str(tmp_path / "out.bin"), str(iters)], capture_output=True, text=True)
    print(p.stderr)
    blob = (tmp_path / "out.bin").read_bytes()
    pos = 0

    def take(dt, n):
        nonlocal pos
        a = np.frombuffer(blob, dt, n, pos)
        pos += a.nbytes
        return a

    # the single-threaded baselines are the oracle's results
    for mono, k, d in ((monoL, kl, dl), (monoR, kr, dr)):
        m, n = take(np.int32, 2)
        assert (m, n) == (mono, len(k))
        assert np.array_equal(take(oracle.KP_DTYPE, n).view(np.uint8), k.view(np.uint8))
        assert np.array_equal(take(np.uint8, 32 * n).reshape(n, 32), d)
    bounds = (0.0, float(cols), 0.0, float(rows))
    rf = oracle.OracleFrame(kl, dl, ur, bounds, scale)
    nm = int(take(np.int32, 1)[0])
    got_mp = take(np.int32, N)
    want_mp = frame_mp0.copy()
    wn = rf.SearchByProjection_mps(mp, want_mp, 3.0, bFarPoints=True, thFarPoints=60.0, nnratio=0.8)
    assert nm == wn and wn > 100 and np.array_equal(got_mp, want_mp)
    bi, bd = take(np.int32, Mf), take(np.int32, Mf)
    wi, wd = rf.FuseSearch(inv_sigma2, fvalid, fu, fv, fur, flevel, fradius, fdesc)
    assert np.array_equal(bi, wi) and np.array_equal(bd, wd) and (wi >= 0).sum() > 300
    bn = int(take(np.int32, 1)[0])
    m12, m21 = take(np.int32, len(bp["desc1"])), take(np.int32, len(bp["desc2"]))
    on, o12, o21 = oracle.search_by_bow(bp["desc1"], bp["desc2"], bp["valid1"], bp["avail2"], bp["fv1"], bp["fv2"], bp["angle1"],
                                        bp["angle2"], 50, True, 0.7, True)
    assert bn == on and on > 100 and np.array_equal(m12, o12) and np.array_equal(m21, o21)
    # and nothing changed under concurrency
    mismatches, errors = take(np.int32, 2)
    calls = int(take(np.int64, 1)[0])
    assert pos == len(blob)
    assert p.returncode == 0 and mismatches == 0 and errors == 0
    assert calls >= 3 * 20, calls                # the three matcher threads really ran beside the 100 x 2 extractions
