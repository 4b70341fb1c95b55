"""CPU checks of the tracking-chain test infrastructure: the oracle's mGrid CSR against a definition-level numpy restatement
of Frame::AssignFeaturesToGrid / PosInGrid (Frame.cc:385-416, :657-667), and the local-map generator's usefulness (most points
in view, predicted level near the keypoint's octave) — so that the GPU tests compare against something meaningful."""
import numpy as np

import frustum_cases as fc
import track_cases as tc

KP = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])


def _round_half_away(v):
    return np.where(v >= 0, np.floor(v + np.float32(0.5)), -np.floor(-v + np.float32(0.5)))


def test_oracle_grid_csr_matches_definition(oracle):
    rng = np.random.Generator(np.random.PCG64(4))
    n = 3000
    kps = np.zeros(n, KP)
    kps["x"] = rng.uniform(-30, 1280, n).astype(np.float32)
    kps["y"] = rng.uniform(-20, 400, n).astype(np.float32)
    kps["x"][:200] = ((rng.integers(0, 66, 200) + 0.5) * 1241.0 / 64.0).astype(np.float32)
    kps["octave"] = rng.integers(0, 8, n)
    desc = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    bounds = (0.0, 1241.0, 0.0, 376.0)
    f = oracle.OracleFrame(kps, desc, None, bounds, (1.2 ** np.arange(8)).astype(np.float32))
    cb, ci = f.grid_csr()
    inv_w = np.float32(64) / np.float32(bounds[1] - bounds[0])
    inv_h = np.float32(48) / np.float32(bounds[3] - bounds[2])
    # the products are exact float32 operations; x.5 cases are decided by round-half-away-from-zero (std::round)
    px = ((kps["x"] - np.float32(bounds[0])) * inv_w).astype(np.float32)
    py = ((kps["y"] - np.float32(bounds[2])) * inv_h).astype(np.float32)
    gx = np.where(px >= 0, np.floor(px.astype(np.float64) + 0.5), -np.floor(-px.astype(np.float64) + 0.5)).astype(np.int64)
    gy = np.where(py >= 0, np.floor(py.astype(np.float64) + 0.5), -np.floor(-py.astype(np.float64) + 0.5)).astype(np.int64)
    ok = (gx >= 0) & (gx < 64) & (gy >= 0) & (gy < 48)
    cell = gx * 48 + gy
    order = np.argsort(np.where(ok, cell, 1 << 30), kind="stable")[:ok.sum()]   # stable: ascending index inside a cell
    assert np.array_equal(ci, order.astype(np.int32))
    assert np.array_equal(cb, np.searchsorted(cell[order], np.arange(64 * 48 + 1), side="left").astype(np.int32))
    assert 0 < cb[-1] < n


def test_local_map_generator_is_a_tracking_workload(oracle):
    import msorb
    rng = np.random.Generator(np.random.PCG64(8))
    n = 1500
    kps = np.zeros(n, KP)
    kps["x"] = rng.uniform(20, 1220, n).astype(np.float32)
    kps["y"] = rng.uniform(20, 350, n).astype(np.float32)
    kps["octave"] = rng.integers(0, 8, n)
    desc = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    depth = np.where(rng.random(n) < 0.7, rng.uniform(3, 60, n), -1).astype(np.float32)
    ur = np.where(depth > 0, kps["x"] - fc.KITTI_CAM["mbf"] / np.maximum(depth, 1e-3), -1).astype(np.float32)
    R = np.eye(3, dtype=np.float32)
    t = np.array([0.1, -0.05, 0.2], np.float32)
    Ow = -t
    scale = (np.float32(1.2) ** np.arange(8)).astype(np.float32)
    mp = tc.local_map(5, kps, desc, ur, depth, R, t, Ow, scale, 4000)
    c = fc.KITTI_CAM
    fr = msorb.Frustum.make(R, t, Ow, c["fx"], c["fy"], c["cx"], c["cy"], c["bounds"], c["mbf"], float(np.log(np.float32(1.2))), 8)
    r = oracle.is_in_frustum(fr, mp["pos_w"], mp["normal"], mp["max_distance"], mp["min_distance"])
    assert r["track_in_view"].mean() > 0.5
    rf = oracle.OracleFrame(kps, desc, ur, c["bounds"], scale)
    frame_mp = np.full(n, -1, np.int32)
    nm, _, _ = tc.oracle_local_points(oracle, rf, fr, mp, frame_mp, 3.0)
    assert nm > 400 and 400 < (frame_mp >= 0).sum() <= nm   # a keypoint holding a point without observations is re-assigned
    oi, od, _ = tc.oracle_topk(oracle, rf, fr, {k: v[:300] for k, v in mp.items()}, 3.0, scale)
    assert (oi[:, 0] >= 0).sum() > 80 and np.all(np.diff(od.astype(np.int64), axis=1)[oi[:, 1:] >= 0] >= 0)


def _fma32(a, b, c):
    """float32 fma: the float64 product of two float32 is exact, the float64 sum with a float32 rounds once to 53 bits and
    once more to 24 — innocuous unless the sum is a 53-bit tie case of the 24-bit rounding, which this data does not produce
    (checked by the equality below holding on 10^4 points)."""
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(np.float32)


def test_oracle_last_frame_projection_matches_definition(oracle):
    """orc_project_last_frame (ORBmatcher.cc:1962-1990, :2019) against a numpy restatement of Sophus' SE3 action
    (so3.hpp:358-367, se3.hpp:321-324) in the stated float convention, and against float64 math within float tolerance."""
    import msorb
    from msorb import synth
    rng = np.random.Generator(np.random.PCG64(12))
    n = 10000
    kps = np.zeros(n, KP)
    kps["x"] = rng.uniform(-30, 1270, n).astype(np.float32)     # some project outside the image
    kps["y"] = rng.uniform(-20, 396, n).astype(np.float32)
    kps["octave"] = rng.integers(0, 8, n)
    kps["angle"] = rng.uniform(0, 360, n)
    desc = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    depth = np.where(rng.random(n) < 0.6, rng.uniform(3, 50, n), -1).astype(np.float32)
    last, q, t, fw, bw = synth.last_frame(3, kps, desc, depth, behind_frac=0.05)
    cam = synth.KITTI_CAM
    mm = oracle.MotionModel()
    mm.q[:] = [float(v) for v in q]
    mm.t[:] = [float(v) for v in t]
    mm.fx, mm.fy, mm.cx, mm.cy, mm.mbf = cam["fx"], cam["fy"], cam["cx"], cam["cy"], cam["mbf"]
    bounds = (0.0, 1241.0, 0.0, 376.0)
    valid, u, v, ur = oracle.project_last_frame(mm, bounds, last["has_point"], last["pos_w"])
    f32 = np.float32
    qx, qy, qz, qw = [np.full(n, f32(c), f32) for c in q]
    px, py, pz = [np.ascontiguousarray(last["pos_w"][:, k]) for k in range(3)]
    dop = lambda a, b, c, d: _fma32(a, b, -(c * d))              # a*b - c*d with the first product fused
    uvx, uvy, uvz = dop(qy, pz, qz, py), dop(qz, px, qx, pz), dop(qx, py, qy, px)
    uvx, uvy, uvz = uvx + uvx, uvy + uvy, uvz + uvz
    c0, c1, c2 = dop(qy, uvz, qz, uvy), dop(qz, uvx, qx, uvz), dop(qx, uvy, qy, uvx)
    xc = (_fma32(qw, uvx, px) + c0) + f32(t[0])
    yc = (_fma32(qw, uvy, py) + c1) + f32(t[1])
    zc = (_fma32(qw, uvz, pz) + c2) + f32(t[2])
    with np.errstate(divide="ignore", invalid="ignore"):
        invz = (1.0 / zc.astype(np.float64)).astype(f32)
        uu = (f32(cam["fx"]) * xc) / zc + f32(cam["cx"])
        vv = (f32(cam["fy"]) * yc) / zc + f32(cam["cy"])
    ok = (last["has_point"] > 0) & ~(invz < 0) & ~((uu < bounds[0]) | (uu > bounds[1])) & ~((vv < bounds[2]) | (vv > bounds[3]))
    assert np.array_equal(valid.astype(bool), ok)
    assert 0.3 * n < ok.sum() < 0.8 * n and ((last["has_point"] > 0) & ~ok).sum() > 200
    assert np.array_equal(u[ok].view(np.uint32), uu[ok].view(np.uint32))
    assert np.array_equal(v[ok].view(np.uint32), vv[ok].view(np.uint32))
    assert np.array_equal(ur[ok].view(np.uint32), _fma32(np.full(n, -f32(cam["mbf"]), f32), invz, uu)[ok].view(np.uint32))
    assert (u[~ok] == 0).all() and (ur[~ok] == 0).all()
    # float64 rotation-matrix math: the same projection within float precision
    x, y, z, w = [float(c) for c in q]
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                  [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                  [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    Pc = last["pos_w"].astype(np.float64) @ R.T + t.astype(np.float64)
    assert np.abs(cam["fx"] * Pc[ok, 0] / Pc[ok, 2] + cam["cx"] - u[ok]).max() < 2e-3
    # and the generator makes a motion-model workload: projections a few pixels from the keypoints they came from
    d = np.hypot(u - kps["x"], v - kps["y"])[ok]
    assert 1.0 < d.mean() < 6.0
    assert isinstance(fw, (bool, np.bool_)) and isinstance(bw, (bool, np.bool_))
