"""Frame::isInFrustum pre-pass: oracle sanity on CPU, device vs oracle bit-exact on the GPU, restated logf."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "ms-slam_amd"), os.path.join(ROOT, "oracle"), os.path.dirname(os.path.abspath(__file__))]
import frustum_cases as fc  # noqa: E402


def _frustum(R, t, Ow, nlevels=8):
    import msorb
    c = fc.KITTI_CAM
    return msorb.Frustum.make(R, t, Ow, c["fx"], c["fy"], c["cx"], c["cy"], c["bounds"], c["mbf"],
                              float(np.log(np.float32(1.2))), nlevels)


def test_restated_logf_matches_installed_glibc(tmp_path):
    exe = tmp_path / "logf_check"
    subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-o", str(exe), os.path.join(ROOT, "tests", "logf_check.cc"), "-lm"])
    stride = "1" if os.environ.get("MSORB_EXHAUSTIVE") else "97"
    out = subprocess.check_output([str(exe), stride]).decode()
    assert "mismatches 0" in out, out


def test_oracle_identity_pose_hand_values():
    import orb_oracle
    R, t, Ow = np.eye(3, dtype=np.float32), np.zeros(3, np.float32), np.zeros(3, np.float32)
    F = _frustum(R, t, Ow)
    c = fc.KITTI_CAM
    P = np.array([[0, 0, 10], [0, 0, -1], [100, 0, 1], [0, 0, 10], [0, 0, 10], [1, 2, 10]], np.float32)
    N = np.array([[0, 0, 1], [0, 0, 1], [0, 0, 1], [0, 0, -1], [0, 0, 1], [0, 0, 1]], np.float32)
    maxd = np.array([10 * 1.2 ** 3, 10, 10, 10, 5, 20], np.float32)
    mind = np.array([1, 1, 1, 1, 1, 1], np.float32)
    r = orb_oracle.is_in_frustum(F, P, N, maxd, mind)
    assert r["track_in_view"].tolist() == [1, 0, 0, 0, 0, 1]
    assert r["proj_x"][0] == np.float32(c["cx"]) and r["proj_y"][0] == np.float32(c["cy"])
    assert r["proj_x"][1] == -1 and r["proj_x"][2] == -1                 # behind camera / outside the image
    assert r["proj_x"][3] == np.float32(c["cx"])                         # projected, then rejected by the viewing angle
    assert r["proj_x"][4] == np.float32(c["cx"])                         # ... by the distance range (10 > 1.2*5)
    assert r["level"][0] == 3 and r["track_depth"][0] == 10 and r["view_cos"][0] == 1
    assert r["proj_xr"][0] == np.float32(np.float32(c["cx"]) - np.float32(c["mbf"]) * np.float32(0.1))
    u5 = np.float32(np.float32(np.float32(c["fx"]) * np.float32(1)) / np.float32(10)) + np.float32(c["cx"])
    assert r["proj_x"][5] == u5


@pytest.mark.gpu
@pytest.mark.parametrize("seed,n", [(0, 1), (1, 1000), (2, 50000), (3, 200001)])
def test_device_matches_oracle(seed, n):
    import msorb
    import orb_oracle
    R, t, Ow = fc.pose(seed)
    F = _frustum(R, t, Ow)
    P, N, maxd, mind = fc.points(seed, n, R, t, Ow)
    a = msorb.is_in_frustum(F, P, N, maxd, mind)
    b = orb_oracle.is_in_frustum(F, P, N, maxd, mind)
    for k in ("track_in_view", "proj_x", "proj_y", "proj_xr", "track_depth", "level", "view_cos"):
        assert a[k].tobytes() == b[k].tobytes(), k
    if n >= 1000:
        frac = a["track_in_view"].mean()
        assert 0.05 < frac < 0.9, frac                                   # the case set exercises both outcomes
        assert len(np.unique(a["level"][a["track_in_view"] > 0])) >= 6


@pytest.mark.gpu
def test_frustum_feeds_search_by_projection():
    """isInFrustum outputs go straight into msorb_search_by_projection_mps (Tracking::SearchLocalPoints chain)."""
    import msorb
    import orb_oracle
    import matcher_cases as mc
    from msorb import synth
    cfg = synth.KITTI
    ex = msorb.ORBextractor(cfg["nfeatures"], cfg["scale"], cfg["nlevels"], cfg["ini_th"], cfg["min_th"])
    try:
        _, kps, desc = ex(synth.image(3, cfg["rows"], cfg["cols"]))
        scale = ex.GetScaleFactors()
    finally:
        ex.close()
    # map points: back-project keypoints at random depths through an identity pose, then perturb
    rng = np.random.default_rng(1)
    c = fc.KITTI_CAM
    M = len(kps)
    z = rng.uniform(4, 60, M).astype(np.float32)
    P = np.stack([(kps["x"] - c["cx"]) * z / c["fx"], (kps["y"] - c["cy"]) * z / c["fy"], z], 1).astype(np.float32)
    P += rng.normal(scale=0.01, size=P.shape).astype(np.float32)
    N = P / np.linalg.norm(P, axis=1, keepdims=True)
    dist = np.linalg.norm(P, axis=1)
    maxd = (dist * scale[kps["octave"]]).astype(np.float32)
    mind = (maxd / scale[-1]).astype(np.float32)
    F = _frustum(np.eye(3, dtype=np.float32), np.zeros(3, np.float32), np.zeros(3, np.float32))
    a = msorb.is_in_frustum(F, P, N, maxd, mind)
    b = orb_oracle.is_in_frustum(F, P, N, maxd, mind)
    assert a["track_in_view"].sum() > 0.9 * M
    flips = rng.integers(0, 30, M)
    mdesc = desc.copy()
    for i in range(M):
        pos = rng.integers(0, 256, flips[i])
        np.bitwise_xor.at(mdesc[i], pos // 8, (1 << (pos % 8)).astype(np.uint8))

    def table(r):
        return dict(track_in_view=r["track_in_view"], bad=np.zeros(M, np.uint8), sparsified=np.zeros(M, np.uint8),
                    proj_x=r["proj_x"], proj_y=r["proj_y"], proj_xr=r["proj_xr"], track_depth=r["track_depth"],
                    level=r["level"], view_cos=r["view_cos"], desc=mdesc, obs=np.ones(M, np.int32))

    bounds = (0.0, float(cfg["cols"]), 0.0, float(cfg["rows"]))
    fr = msorb.Frame(kps, desc, None, bounds, scale)
    orf = orb_oracle.OracleFrame(kps, desc, None, bounds, scale)
    try:
        fm_a, fm_b = np.full(M, -1, np.int32), np.full(M, -1, np.int32)
        na = fr.SearchByProjection_mps(table(a), fm_a, 3.0)
        nb = orf.SearchByProjection_mps(table(b), fm_b, 3.0)
        assert na == nb and na > 0.5 * M
        assert fm_a.tolist() == fm_b.tolist()
    finally:
        fr.close()
