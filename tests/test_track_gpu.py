"""GPU parity tests of the tracking front-end chain (csrc/track.hip; BASELINE configs[2]): the device
AssignFeaturesToGrid, the fused isInFrustum + SearchByProjection entry, the one-call frame front-end and its batched
form, all through the C ABI against the CPU oracle.  Index work: everything must be identical."""
import numpy as np
import pytest

from msorb import synth
import frustum_cases as fc
import matcher_cases as mc
import track_cases as tc

pytestmark = pytest.mark.gpu

MBF, MB = mc.KITTI_BF, mc.KITTI_BF / mc.KITTI_FX
BOUNDS = (0.0, 1241.0, 0.0, 376.0)


def _frustum(msorb_mod, R, t, Ow, nlevels=8):
    c = fc.KITTI_CAM
    return msorb_mod.Frustum.make(R, t, Ow, c["fx"], c["fy"], c["cx"], c["cy"], c["bounds"], c["mbf"], float(np.log(np.float32(1.2))), nlevels)


def _near_identity_pose(seed):
    """A pose close to the identity so that most of a synthetic local map lands inside the image."""
    rng = np.random.default_rng(seed)
    w = rng.normal(scale=0.02, size=3)
    th = np.linalg.norm(w) + 1e-12
    k = w / th
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    R = (np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K).astype(np.float32)
    t = rng.normal(scale=0.3, size=3).astype(np.float32)
    Ow = (-(R.T.astype(np.float64) @ t.astype(np.float64))).astype(np.float32)
    return R, t, Ow


@pytest.fixture(scope="module")
def kitti_frame(msorb_mod, oracle):
    cfg = synth.KITTI
    L, R = synth.stereo_pair(21, cfg["rows"], cfg["cols"])
    ex = msorb_mod.ORBextractor(2000, 1.2, 8, 20, 7)
    kl, dl, kr, dr, ur, dp, oob = ex.extract_stereo(L, R, MB, MBF)
    return dict(L=L, R=R, ex=ex, kl=kl, dl=dl, kr=kr, dr=dr, ur=ur, dp=dp, oob=oob, scale=ex.GetScaleFactors())


def _grid_equal(msorb_mod, oracle, kps, desc, ur, bounds, scale):
    f = msorb_mod.Frame(kps, desc, ur, bounds, scale)
    rf = oracle.OracleFrame(kps, desc, ur, bounds, scale)
    cb, ci = msorb_mod.frame_grid(f)
    rcb, rci = rf.grid_csr()
    assert np.array_equal(cb, rcb), "cell_begin differs"
    assert np.array_equal(ci, rci), "cell contents / order differ"
    return f, rf, cb


def test_device_grid_matches_oracle_on_extracted_frames(msorb_mod, oracle, kitti_frame):
    s = kitti_frame
    _, _, cb = _grid_equal(msorb_mod, oracle, s["kl"], s["dl"], s["ur"], BOUNDS, s["scale"])
    assert cb[-1] == len(s["kl"]) > 1500
    # other image bounds (undistorted fisheye-style bounds are not the image size, Frame.cc:144-160)
    _grid_equal(msorb_mod, oracle, s["kl"], s["dl"], None, (-13.5, 1260.25, -7.75, 390.0), s["scale"])
    _grid_equal(msorb_mod, oracle, s["kr"], s["dr"], None, BOUNDS, s["scale"])


@pytest.mark.parametrize("seed,n", [(1, 0), (2, 1), (3, 64), (4, 2000), (5, 5000), (6, 16000), (7, 20000), (8, 32768)])
def test_device_grid_random_keypoints(msorb_mod, oracle, seed, n):
    """Keypoints anywhere, including outside the bounds, exactly on cell borders (x.5 products), and piled up in a few cells
    (long runs exercise the in-cell order restoration)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    kps = np.zeros(n, msorb_mod.KP_DTYPE)
    x = rng.uniform(-40, 1290, n)
    y = rng.uniform(-30, 410, n)
    pile = rng.random(n) < 0.3
    x[pile] = rng.choice([100.0, 600.5, 1239.0], int(pile.sum())) + rng.uniform(-3, 3, int(pile.sum()))
    y[pile] = rng.choice([50.0, 200.0, 370.0], int(pile.sum())) + rng.uniform(-2, 2, int(pile.sum()))
    on_border = rng.random(n) < 0.2   # (x - min) * 64 / 1241 == k + 0.5 -> round() half away from zero decides the cell
    kb = rng.integers(0, 66, n)
    x[on_border] = ((kb[on_border] + 0.5) * 1241.0 / 64.0)
    kps["x"], kps["y"] = x.astype(np.float32), y.astype(np.float32)
    kps["octave"] = rng.integers(0, 8, n)
    kps["size"], kps["angle"] = 31.0, rng.uniform(0, 360, n)
    desc = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    ur = np.where(rng.random(n) < 0.5, x - rng.uniform(1, 40, n), -1).astype(np.float32)
    scale = (1.2 ** np.arange(8)).astype(np.float32)
    f, rf, cb = _grid_equal(msorb_mod, oracle, kps, desc, ur, BOUNDS, scale)
    for _ in range(40):   # the host walk reads the grid it fetched from the device
        qx, qy, r = rng.uniform(-50, 1300), rng.uniform(-50, 420), rng.uniform(1, 60)
        assert np.array_equal(f.GetFeaturesInArea(qx, qy, r, 0, 3), rf.GetFeaturesInArea(qx, qy, r, 0, 3))


def test_frame_from_device_arrays(msorb_mod, oracle, kitti_frame):
    import torch
    s = kitti_frame
    n = len(s["kl"])
    d_kps = torch.from_numpy(np.frombuffer(s["kl"].tobytes(), np.uint8).reshape(n, 28).copy()).cuda()
    d_desc = torch.from_numpy(s["dl"].copy()).cuda()
    d_ur = torch.from_numpy(s["ur"].copy()).cuda()
    f = msorb_mod.frame_from_device(d_kps, n, d_desc, d_ur, BOUNDS, s["scale"])
    rf = oracle.OracleFrame(s["kl"], s["dl"], s["ur"], BOUNDS, s["scale"])
    cb, ci = msorb_mod.frame_grid(f)
    rcb, rci = rf.grid_csr()
    assert np.array_equal(cb, rcb) and np.array_equal(ci, rci)
    rng = np.random.Generator(np.random.PCG64(9))
    mp = mc.map_point_table(rng, s["kl"], s["dl"], s["ur"], s["scale"], 3000)
    a, b = np.full(n, -1, np.int32), np.full(n, -1, np.int32)
    assert f.SearchByProjection_mps(mp, a, 3.0) == rf.SearchByProjection_mps(mp, b, 3.0) > 300
    assert np.array_equal(a, b)


def test_extract_stereo_frame_equals_separate_calls(msorb_mod, oracle, kitti_frame):
    s = kitti_frame
    f, (kl, dl, kr, dr, ur, dp, oob) = msorb_mod.extract_stereo_frame(s["ex"], s["L"], s["R"], MB, MBF)
    for a, b in ((kl, s["kl"]), (kr, s["kr"])):
        assert np.array_equal(a.view(np.uint8), b.view(np.uint8))
    assert np.array_equal(dl, s["dl"]) and np.array_equal(dr, s["dr"]) and oob == s["oob"]
    assert np.array_equal(ur.view(np.uint32), s["ur"].view(np.uint32)) and np.array_equal(dp.view(np.uint32), s["dp"].view(np.uint32))
    rf = oracle.OracleFrame(kl, dl, ur, BOUNDS, s["scale"])
    cb, ci = msorb_mod.frame_grid(f)
    rcb, rci = rf.grid_csr()
    assert np.array_equal(cb, rcb) and np.array_equal(ci, rci)
    # the frame serves the other searches like one loaded from host arrays
    rng = np.random.Generator(np.random.PCG64(3))
    last = mc.last_frame_table(rng, kl, dl, ur, s["scale"], 1800)
    a, b = np.full(len(kl), -1, np.int32), np.full(len(kl), -1, np.int32)
    assert f.SearchByProjection_frames(last, a, 7.0) == rf.SearchByProjection_frames(last, b, 7.0) > 200
    assert np.array_equal(a, b)


@pytest.mark.parametrize("seed,M,th,far,pre", [(1, 4096, 1.0, False, 0.0), (2, 4096, 3.0, True, 0.2), (3, 8000, 5.0, False, 0.5),
                                               (4, 700, 1.0, True, 0.0), (5, 0, 1.0, False, 0.0)])
def test_search_local_points_vs_oracle(msorb_mod, oracle, kitti_frame, seed, M, th, far, pre):
    s = kitti_frame
    n = len(s["kl"])
    R, t, Ow = _near_identity_pose(seed)
    fr = _frustum(msorb_mod, R, t, Ow)
    mp = tc.local_map(seed, s["kl"], s["dl"], s["ur"], s["dp"], R, t, Ow, s["scale"], M)
    f = msorb_mod.Frame(s["kl"], s["dl"], s["ur"], BOUNDS, s["scale"])
    rf = oracle.OracleFrame(s["kl"], s["dl"], s["ur"], BOUNDS, s["scale"])
    rng = np.random.Generator(np.random.PCG64(100 + seed))
    init = np.where(rng.random(n) < pre, rng.integers(0, max(M, 1), n), -1).astype(np.int32) if M else np.full(n, -1, np.int32)
    a, b = init.copy(), init.copy()
    nm, out = msorb_mod.search_local_points(f, fr, mp, a, th, far, 40.0)
    rnm, r, visit = tc.oracle_local_points(oracle, rf, fr, mp, b, th, far, 40.0)
    assert nm == rnm
    assert np.array_equal(a, b)
    if M:
        assert nm > M // 20
        assert np.array_equal(out["track_in_view"].astype(bool), r["track_in_view"].astype(bool) & visit)
        v = visit & r["track_in_view"].astype(bool)
        for k in ("proj_x", "proj_y", "proj_xr", "track_depth", "view_cos"):
            assert np.array_equal(out[k][v].view(np.uint32), r[k][v].view(np.uint32)), k
        assert np.array_equal(out["level"][v], r["level"][v])
    # identical to the two separate entries
    r2 = msorb_mod.is_in_frustum(fr, mp["pos_w"], mp["normal"], mp["max_distance"], mp["min_distance"])
    tab = dict(track_in_view=(r2["track_in_view"].astype(bool) & visit).astype(np.uint8), bad=mp["bad"], sparsified=mp["sparsified"],
               proj_x=r2["proj_x"], proj_y=r2["proj_y"], proj_xr=r2["proj_xr"], track_depth=r2["track_depth"], level=r2["level"],
               view_cos=r2["view_cos"], desc=mp["desc"], obs=mp["obs"])
    c = init.copy()
    assert f.SearchByProjection_mps(tab, c, th, far, 40.0) == nm and np.array_equal(c, a)


@pytest.mark.parametrize("seed,M,th", [(11, 4096, 1.0), (12, 6000, 3.0)])
def test_track_frontend_one_call(msorb_mod, oracle, kitti_frame, seed, M, th):
    s = kitti_frame
    R, t, Ow = _near_identity_pose(seed)
    fr = _frustum(msorb_mod, R, t, Ow)
    mp = tc.local_map(seed, s["kl"], s["dl"], s["ur"], s["dp"], R, t, Ow, s["scale"], M)
    f, st, frame_mp, nm, out, rounds = msorb_mod.track_frontend(s["ex"], s["L"], s["R"], MB, MBF, fr, mp, th)
    kl, dl, kr, dr, ur, dp, oob = st
    assert np.array_equal(kl.view(np.uint8), s["kl"].view(np.uint8)) and np.array_equal(dl, s["dl"])
    assert np.array_equal(ur.view(np.uint32), s["ur"].view(np.uint32)) and np.array_equal(dp.view(np.uint32), s["dp"].view(np.uint32))
    rf = oracle.OracleFrame(kl, dl, ur, BOUNDS, s["scale"])
    b = np.full(len(kl), -1, np.int32)
    rnm, r, visit = tc.oracle_local_points(oracle, rf, fr, mp, b, th)
    assert nm == rnm > M // 20 and np.array_equal(frame_mp, b)
    assert rounds >= 1
    assert np.array_equal(out["track_in_view"].astype(bool), r["track_in_view"].astype(bool) & visit)
    # the frame left behind serves the next search (TrackWithMotionModel of the next frame projects INTO a new frame, but
    # relocalisation / SearchByProjection(F, KF) run against this one)
    cb, ci = msorb_mod.frame_grid(f)
    rcb, rci = rf.grid_csr()
    assert np.array_equal(cb, rcb) and np.array_equal(ci, rci)


def test_track_frontend_many_points_per_keypoint_forces_rounds(msorb_mod, oracle, kitti_frame):
    """20 000 map points on 2000 keypoints: candidate lists get exhausted by earlier claims, the replay re-runs the kernel."""
    s = kitti_frame
    R, t, Ow = _near_identity_pose(31)
    fr = _frustum(msorb_mod, R, t, Ow)
    mp = tc.local_map(31, s["kl"], s["dl"], s["ur"], s["dp"], R, t, Ow, s["scale"], 20000, copy_frac=0.95, obs_zero_frac=0.0,
                      spars_frac=0.0)
    f, st, frame_mp, nm, out, rounds = msorb_mod.track_frontend(s["ex"], s["L"], s["R"], MB, MBF, fr, mp, 3.0)
    rf = oracle.OracleFrame(st[0], st[1], st[4], BOUNDS, s["scale"])
    b = np.full(len(st[0]), -1, np.int32)
    rnm, _, _ = tc.oracle_local_points(oracle, rf, fr, mp, b, 3.0)
    assert nm == rnm and np.array_equal(frame_mp, b)
    last, total, searches = msorb_mod.frame_search_rounds(f)
    assert last == rounds >= 2 and total >= last and searches >= 1      # lists were exhausted: the kernel was re-run


@pytest.mark.parametrize("M,th", [(1500, 3.0), (11000, 1.0)])
def test_track_batch_vs_oracle(msorb_mod, oracle, M, th):
    """(11000 points x 3 frames at th = 1 is past the threshold where the window kernel runs 4 lanes per query.)"""
    import torch
    cfg = synth.KITTI
    n_pairs = 3
    imgs = []
    for p in range(n_pairs):
        L, R = synth.stereo_pair(40 + p, cfg["rows"], cfg["cols"])
        imgs += [L, R]
    ex = msorb_mod.ORBextractor(2000, 1.2, 8, 20, 7)
    d_img = torch.from_numpy(np.stack(imgs)).cuda()
    counts, mono, d_kps, d_desc = ex.extract_batch(d_img)
    d_ur, d_dp, oob, _ = msorb_mod.stereo_matches_batch(ex, counts, d_kps, d_desc, MB, MBF)
    scale = ex.GetScaleFactors()
    kps_all = msorb_mod.keypoints_from_device(d_kps, counts)
    desc_all = d_desc.cpu().numpy()
    ur_all, dp_all = d_ur.cpu().numpy(), d_dp.cpu().numpy()
    frusta, mps, poses = [], [], []
    for p in range(n_pairs):
        R, t, Ow = _near_identity_pose(60 + p)
        n = counts[2 * p]
        mp = tc.local_map(70 + p, kps_all[2 * p], desc_all[2 * p][:n], ur_all[p][:n], dp_all[p][:n], R, t, Ow, scale, M)
        mp["flags"] = (mp["visit"] | (mp["bad"] << 1) | (mp["sparsified"] << 2)).astype(np.uint8)
        frusta.append(_frustum(msorb_mod, R, t, Ow))
        mps.append(mp)
    d_mp = {k: torch.from_numpy(np.stack([m[k] for m in mps])).cuda().contiguous()
            for k in ("pos_w", "normal", "max_distance", "min_distance", "flags", "desc")}
    r = msorb_mod.track_batch(d_kps, d_desc, d_ur, counts, 2, BOUNDS, scale, frusta, d_mp, th, want_grid=True, count_pairs=True)
    ti, td = r["topk_idx"].cpu().numpy(), r["topk_dist"].cpu().numpy()
    cb, ci = r["cell_begin"].cpu().numpy(), r["cell_idx"].cpu().numpy()
    n_eval = 0
    for p in range(n_pairs):
        n = counts[2 * p]
        rf = oracle.OracleFrame(kps_all[2 * p], desc_all[2 * p][:n], ur_all[p][:n], BOUNDS, scale)
        rcb, rci = rf.grid_csr()
        assert np.array_equal(cb[p], rcb) and np.array_equal(ci[p][:len(rci)], rci)
        oi, od, fr_out = tc.oracle_topk(oracle, rf, frusta[p], mps[p], th, scale)
        assert np.array_equal(ti[p], oi), f"pair {p}: candidate indices differ"
        assert np.array_equal(td[p][oi >= 0], od[oi >= 0])
        assert (oi[:, 0] >= 0).sum() > M // 6
        inv = r["in_view"][p].cpu().numpy().astype(bool)
        assert np.array_equal(inv, fr_out["track_in_view"].astype(bool) & mps[p]["visit"].astype(bool))
    assert r["n_pairs"] > 0 and all(m >= 0 for m in r["ms"])
    # without the counter the lists are the same
    r2 = msorb_mod.track_batch(d_kps, d_desc, d_ur, counts, 2, BOUNDS, scale, frusta, d_mp, th)
    assert torch.equal(r2["topk_idx"], r["topk_idx"]) and torch.equal(r2["topk_dist"], r["topk_dist"])


# ----------------------------------------------------------------------------------------------------------------------
# TrackWithMotionModel's search (a14) with the projection on the device: msorb_frame_set_last_points +
# msorb_search_last_frame / msorb_track_frontend_motion vs the oracle composition orc_project_last_frame +
# orc_search_by_projection_frames (ORBmatcher.cc:1941-2152; caller Tracking.cc:2833-2870)
# ----------------------------------------------------------------------------------------------------------------------
def _motion_model(mod, q, t, forward=False, backward=False):
    c = synth.KITTI_CAM
    m = mod.MotionModel()
    m.q[:] = [float(x) for x in q]
    m.t[:] = [float(x) for x in t]
    m.fx, m.fy, m.cx, m.cy, m.mbf = c["fx"], c["fy"], c["cx"], c["cy"], c["mbf"]
    m.forward, m.backward = int(forward), int(backward)
    return m


def _oracle_last_frame(oracle, rf, last, q, t, forward, backward, cur_mp, th, check_orientation=True, bounds=BOUNDS):
    omm = _motion_model(oracle, q, t, forward, backward)
    valid, u, v, ur = oracle.project_last_frame(omm, bounds, last["has_point"], last["pos_w"])
    n = len(valid)
    tab = dict(valid=valid, u=u, v=v, ur=ur, octave=last["octave"], angle=last["angle"], desc=last["desc"],
               mp=np.arange(n, dtype=np.int32), obs=last["obs"])
    nm = rf.SearchByProjection_frames(tab, cur_mp, th, forward, backward, check_orientation)
    return nm, dict(valid=valid, u=u, v=v, ur=ur)


@pytest.mark.parametrize("th,mode", [(7.0, "none"), (15.0, "none"), (7.0, "forward"), (7.0, "backward"), (15.0, "mono"),
                                     (14.0, "none"), (30.0, "forward")])
def test_search_last_frame_device_projection(msorb_mod, oracle, kitti_frame, th, mode):
    """th = 7 (stereo) / 15 (monocular) and the retry at 2 * th (Tracking.cc:2845-2868); forward / backward level bands
    (:1993-1998); mono = a frame without mvuRight (no :2015-2022 test, bForward = bBackward = false)."""
    s = kitti_frame
    ur = None if mode == "mono" else s["ur"]
    depth = np.full(len(s["kl"]), -1.0, np.float32) if mode == "mono" else s["dp"]
    last, q, t, _, _ = synth.last_frame(100 + int(th), s["kl"], s["dl"], depth)
    fw, bw = mode == "forward", mode == "backward"
    f = msorb_mod.Frame(s["kl"], s["dl"], ur, BOUNDS, s["scale"])
    rf = oracle.OracleFrame(s["kl"], s["dl"], ur, BOUNDS, s["scale"])
    try:
        mm = _motion_model(msorb_mod, q, t, fw, bw)
        msorb_mod.frame_set_last_points(f, last)
        cur = np.full(len(s["kl"]), -1, np.int32)
        nm, proj = msorb_mod.search_last_frame(f, mm, last["obs"], cur, th, True, want_projection=True)
        want = np.full(len(s["kl"]), -1, np.int32)
        wnm, wproj = _oracle_last_frame(oracle, rf, last, q, t, fw, bw, want, th)
        for k in ("valid", "u", "v", "ur"):
            a, b = proj[k], wproj[k]
            assert np.array_equal(a.view(np.uint8), b.view(np.uint8)), k    # bit patterns of the projections
        assert wnm > 300 and (wproj["valid"] == 0).sum() > 100
        assert nm == wnm and np.array_equal(cur, want)
        # the retry of Tracking.cc:2861-2868 on the table already resident: clear, search again at 2 * th
        cur[:] = -1
        want[:] = -1
        nm2 = msorb_mod.search_last_frame(f, mm, last["obs"], cur, 2 * th, True)
        wnm2, _ = _oracle_last_frame(oracle, rf, last, q, t, fw, bw, want, 2 * th)
        assert nm2 == wnm2 and np.array_equal(cur, want)
        # without the orientation check
        cur[:] = -1
        want[:] = -1
        nm3 = msorb_mod.search_last_frame(f, mm, last["obs"], cur, th, False)
        wnm3, _ = _oracle_last_frame(oracle, rf, last, q, t, fw, bw, want, th, False)
        assert nm3 == wnm3 >= nm and np.array_equal(cur, want)
    finally:
        f.close()


def test_search_last_frame_with_points_already_in_the_frame(msorb_mod, oracle, kitti_frame):
    """The general form of the signature: the current frame already holds map points (ids >= n of the table); those with
    observations are never overwritten (:2011-2013), the others may be."""
    s = kitti_frame
    n = len(s["kl"])
    last, q, t, _, _ = synth.last_frame(321, s["kl"], s["dl"], s["dp"], pixel_sigma=1.5)
    rng = np.random.default_rng(5)
    held = rng.random(n) < 0.3
    extra_obs = np.where(rng.random(n) < 0.4, 0, rng.integers(1, 9, n)).astype(np.int32)
    obs = np.concatenate([last["obs"], extra_obs])
    cur0 = np.where(held, n + np.arange(n), -1).astype(np.int32)
    f = msorb_mod.Frame(s["kl"], s["dl"], s["ur"], BOUNDS, s["scale"])
    rf = oracle.OracleFrame(s["kl"], s["dl"], s["ur"], BOUNDS, s["scale"])
    try:
        mm = _motion_model(msorb_mod, q, t)
        msorb_mod.frame_set_last_points(f, last)
        cur = cur0.copy()
        nm = msorb_mod.search_last_frame(f, mm, obs, cur, 7.0, True)
        omm = _motion_model(oracle, q, t)
        valid, u, v, ur = oracle.project_last_frame(omm, BOUNDS, last["has_point"], last["pos_w"])
        tab = dict(valid=valid, u=u, v=v, ur=ur, octave=last["octave"], angle=last["angle"], desc=last["desc"],
                   mp=np.arange(n, dtype=np.int32), obs=obs)
        want = cur0.copy()
        wnm = rf.SearchByProjection_frames(tab, want, 7.0, False, False, True)
        assert nm == wnm and np.array_equal(cur, want)
        assert (want[held & (extra_obs > 0)] >= n).all()          # protected keypoints kept their points
        assert ((want >= 0) & (want < n)).sum() > 100
    finally:
        f.close()


def test_search_last_frame_degenerate_tables(msorb_mod, oracle, kitti_frame):
    s = kitti_frame
    n = len(s["kl"])
    f = msorb_mod.Frame(s["kl"], s["dl"], s["ur"], BOUNDS, s["scale"])
    try:
        mm = _motion_model(msorb_mod, [0, 0, 0, 1], [0, 0, 0])
        cur = np.full(n, -1, np.int32)
        with pytest.raises(msorb_mod.MsorbError):                 # no table yet
            msorb_mod.search_last_frame(f, mm, np.zeros(0, np.int32), cur, 7.0)
        empty = dict(has_point=np.zeros(0, np.uint8), pos_w=np.zeros((0, 3), np.float32), octave=np.zeros(0, np.int32),
                     angle=np.zeros(0, np.float32), desc=np.zeros((0, 32), np.uint8), obs=np.zeros(0, np.int32))
        msorb_mod.frame_set_last_points(f, empty)
        assert msorb_mod.search_last_frame(f, mm, empty["obs"], cur, 7.0) == 0 and (cur == -1).all()
        # nobody holds a point / everything behind the camera
        last, q, t, _, _ = synth.last_frame(9, s["kl"], s["dl"], s["dp"], point_frac=0.0)
        msorb_mod.frame_set_last_points(f, last)
        assert msorb_mod.search_last_frame(f, _motion_model(msorb_mod, q, t), last["obs"], cur, 7.0) == 0
        last, q, t, _, _ = synth.last_frame(9, s["kl"], s["dl"], s["dp"], behind_frac=1.0)
        msorb_mod.frame_set_last_points(f, last)
        nm, proj = msorb_mod.search_last_frame(f, _motion_model(msorb_mod, q, t), last["obs"], cur, 7.0, want_projection=True)
        assert nm == 0 and proj["valid"].sum() == 0
        bad = dict(last, octave=np.full(n, 99, np.int32))
        with pytest.raises(msorb_mod.MsorbError):
            msorb_mod.frame_set_last_points(f, bad)
    finally:
        f.close()


@pytest.mark.parametrize("seed", [21, 3])
def test_track_frontend_motion_equals_oracle_composition(msorb_mod, oracle, seed):
    """Frame::Frame + TrackWithMotionModel's SearchByProjection as ONE call from host images; the same as separate calls; both
    against extraction -> ComputeStereoMatches -> oracle projection -> oracle search."""
    cfg = synth.KITTI
    L, R = synth.stereo_pair(seed, cfg["rows"], cfg["cols"])
    ex = msorb_mod.ORBextractor(2000, 1.2, 8, 20, 7)
    try:
        kl, dl, kr, dr, ur, dp, oob = ex.extract_stereo(L, R, MB, MBF)
        scale = ex.GetScaleFactors()
        last, q, t, _, _ = synth.last_frame(700 + seed, kl, dl, dp)
        mm = _motion_model(msorb_mod, q, t)
        f, st, cur, nm = msorb_mod.track_frontend_motion(ex, L, R, MB, MBF, mm, last, last["obs"], 7.0)
        try:
            assert np.array_equal(st[0].view(np.uint8), kl.view(np.uint8)) and np.array_equal(st[1], dl)
            assert np.array_equal(st[4].view(np.uint32), ur.view(np.uint32))
            rf = oracle.OracleFrame(kl, dl, ur, BOUNDS, scale)
            want = np.full(len(kl), -1, np.int32)
            wnm, _ = _oracle_last_frame(oracle, rf, last, q, t, False, False, want, 7.0)
            assert wnm > 300
            assert nm == wnm and np.array_equal(cur, want)
            # the frame handle serves the follow-up searches: retry at 2 * th on the resident table
            cur2 = np.full(len(kl), -1, np.int32)
            want[:] = -1
            nm2 = msorb_mod.search_last_frame(f, mm, last["obs"], cur2, 14.0)
            wnm2, _ = _oracle_last_frame(oracle, rf, last, q, t, False, False, want, 14.0)
            assert nm2 == wnm2 and np.array_equal(cur2, want)
        finally:
            f.close()
        r = msorb_mod.MotionFrontendRunner(ex, L, R, MB, MBF, mm, last, last["obs"], 7.0)
        try:
            a = r.one_call(); ca = r.cur_mp[:len(kl)].copy()
            b = r.separate_calls(); cb = r.cur_mp[:len(kl)].copy()
            c = r.one_call(); cc = r.cur_mp[:len(kl)].copy()
            assert a == b == c == nm and np.array_equal(ca, cur) and np.array_equal(cb, cur) and np.array_equal(cc, cur)
        finally:
            r.close()
    finally:
        ex.close()


def test_frame_set_beyond_the_grid_capacity_leaves_an_empty_handle(msorb_mod, oracle, kitti_frame):
    """A refused msorb_frame_set (more keypoints than the device grid takes) must not leave the handle half new / half old:
    the call fails before anything is touched, and a handle whose set failed half way is EMPTY — area queries and searches
    find nothing instead of indexing stale arrays."""
    s = kitti_frame
    f = msorb_mod.Frame(s["kl"], s["dl"], s["ur"], BOUNDS, s["scale"])
    try:
        before = f.GetFeaturesInArea(600.0, 180.0, 60.0)
        assert len(before) > 5
        n = 40000
        kps = np.zeros(n, msorb_mod.KP_DTYPE)
        kps["x"], kps["y"] = 10.0, 10.0
        desc = np.zeros((n, 32), np.uint8)
        L = msorb_mod._mlib()
        sf = np.asarray(s["scale"], np.float32)
        rc = L.msorb_frame_set(f.h, msorb_mod._np_ptr(kps), n, msorb_mod._np_ptr(desc), None, BOUNDS[0], BOUNDS[1], BOUNDS[2], BOUNDS[3],
                               msorb_mod._np_ptr(sf), len(sf))
        assert rc == msorb_mod.E_CAPACITY
        # refused up front: the previous frame is still there, intact
        assert f.GetFeaturesInArea(600.0, 180.0, 60.0).tolist() == before.tolist()
    finally:
        f.close()
