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
