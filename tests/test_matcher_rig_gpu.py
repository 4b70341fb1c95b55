"""The two-camera (F.Nleft != -1, KannalaBrandt8 stereo rig) arms of the matcher — ORBmatcher.cc:43-213 (SearchByProjection(F,
vpMapPoints) with its right-camera pass, :144-210) — on the device against the oracle's statement-by-statement arm
(oracle/matcher_oracle.cc orc_search_by_projection_mps_rig).  Synthetic rigs: the right camera sees a shifted, re-detected copy of
part of the left camera's keypoints (mvLeftToRightMatch / mvRightToLeftMatch), several map points per keypoint so that claims
collide on both sides and across them, points seen by one camera only, points whose left ratio test fails (their right pass must
be skipped), occupied keypoints with and without observations, frames with an empty camera."""
import numpy as np
import pytest

import matcher_cases as mc

pytestmark = pytest.mark.gpu
SCALE = np.array([1.2 ** i for i in range(8)], np.float32)
BOUNDS = (0.0, 1241.0, 0.0, 376.0)


def make_rig(oracle, seed, n_left, n_right, M, dense=False, th=1.0):
    rng = np.random.Generator(np.random.PCG64(seed))
    span = (300, 120) if dense else (1200, 340)

    def cam(n):
        k = np.zeros(n, oracle.KP_DTYPE)
        k["x"] = 20 + rng.uniform(0, span[0], n); k["y"] = 20 + rng.uniform(0, span[1], n)
        k["octave"] = rng.integers(0, 8, n); k["angle"] = rng.uniform(0, 360, n); k["size"] = 31
        return k, rng.integers(0, 256, (n, 32), dtype=np.uint8)
    kl, dl = cam(n_left)
    kr, dr = cam(n_right)
    # stereo partners: right keypoint j is the re-detected left keypoint l2r^-1(j)
    l2r = np.full(n_left, -1, np.int32)
    r2l = np.full(n_right, -1, np.int32)
    n_pairs = min(n_left, n_right) * 6 // 10
    li, ri = rng.permutation(n_left)[:n_pairs], rng.permutation(n_right)[:n_pairs]
    l2r[li] = ri; r2l[ri] = li
    if n_pairs:
        dr[ri] = mc.flip_bits(rng, dl[li], 25)
        kr["octave"][ri] = kl["octave"][li]
    # map points: noisy copies of keypoints of either camera, projected near them in both cameras where a partner exists
    from_left = rng.random(M) < 0.6 if n_left and n_right else np.full(M, n_left > 0)
    sl = rng.integers(0, max(n_left, 1), M); sr = rng.integers(0, max(n_right, 1), M)
    has_partner_l = (l2r[sl] >= 0) if n_left else np.zeros(M, bool)
    has_partner_r = (r2l[sr] >= 0) if n_right else np.zeros(M, bool)
    src_l = np.where(from_left, sl, np.where(has_partner_r, r2l[sr] if n_right else 0, sl)) if n_left else np.zeros(M, np.int64)
    src_r = np.where(~from_left, sr, np.where(has_partner_l, l2r[sl] if n_left else 0, sr)) if n_right else np.zeros(M, np.int64)
    base = np.where(from_left[:, None], dl[src_l] if n_left else 0, dr[src_r] if n_right else 0).astype(np.uint8)
    desc = np.where((rng.random(M) < 0.85)[:, None], mc.flip_bits(rng, base, 45), rng.integers(0, 256, (M, 32), dtype=np.uint8)).astype(np.uint8)
    in_l = (rng.random(M) < 0.85) & (n_left > 0) & (from_left | has_partner_r)
    in_r = (rng.random(M) < 0.85) & (n_right > 0) & (~from_left | has_partner_l)
    lvl_l = np.clip((kl["octave"][src_l] if n_left else 0) + rng.integers(0, 2, M), 0, 7).astype(np.int32)
    lvl_r = np.clip((kr["octave"][src_r] if n_right else 0) + rng.integers(0, 2, M), 0, 7).astype(np.int32)
    lvl_r[rng.random(M) < 0.05] = -1       # mnTrackScaleLevelR == -1: no right pass (:146)
    mp = dict(track_in_view=in_l.astype(np.uint8), track_in_view_r=in_r.astype(np.uint8), bad=(rng.random(M) < 0.03).astype(np.uint8),
              sparsified=(rng.random(M) < 0.06).astype(np.uint8),
              proj_x=((kl["x"][src_l] if n_left else np.zeros(M)) + rng.normal(0, 2.5, M)).astype(np.float32),
              proj_y=((kl["y"][src_l] if n_left else np.zeros(M)) + rng.normal(0, 2.5, M)).astype(np.float32),
              proj_xr=((kr["x"][src_r] if n_right else np.zeros(M)) + rng.normal(0, 2.5, M)).astype(np.float32),
              proj_yr=((kr["y"][src_r] if n_right else np.zeros(M)) + rng.normal(0, 2.5, M)).astype(np.float32),
              track_depth=rng.uniform(2, 80, M).astype(np.float32), level=lvl_l, level_r=lvl_r,
              view_cos=rng.uniform(0.99, 1.0, M).astype(np.float32), view_cos_r=rng.uniform(0.99, 1.0, M).astype(np.float32), desc=desc,
              obs=np.where(rng.random(M) < 0.15, 0, rng.integers(1, 12, M)).astype(np.int32))
    frame_mp = np.where(rng.random(n_left + n_right) < 0.15, rng.integers(0, max(M, 1), n_left + n_right), -1).astype(np.int32)
    if M == 0:
        frame_mp[:] = -1
    return dict(kl=kl, dl=dl, kr=kr, dr=dr, l2r=l2r, r2l=r2l, mp=mp, frame_mp=frame_mp, th=th)


CASES = [dict(seed=1, n_left=1500, n_right=1400, M=4096), dict(seed=2, n_left=900, n_right=1100, M=3000, dense=True, th=3.0),
         dict(seed=3, n_left=2000, n_right=1, M=2500), dict(seed=4, n_left=0, n_right=800, M=1500), dict(seed=5, n_left=700, n_right=0, M=1500),
         dict(seed=6, n_left=1200, n_right=1200, M=0), dict(seed=7, n_left=400, n_right=400, M=6000, dense=True, th=4.0),
         dict(seed=8, n_left=1800, n_right=1700, M=5000, th=2.0),
         # found by tests/_fuzz_matcher.py: the FIRST map point has no observations, its left match overwrites the right partner's occupied slot
         # (:130-134) — which frees it for the point's own right pass, between that camera's device round and its first query
         dict(seed=674330, n_left=2062, n_right=2278, M=5219, dense=True, th=2.5111543772514895)]


@pytest.mark.parametrize("case", CASES, ids=lambda c: f"seed{c['seed']}")
def test_search_by_projection_two_camera_frame(msorb_mod, oracle, case):
    R = make_rig(oracle, **case)
    ofl, ofr = oracle.OracleFrame(R["kl"], R["dl"], None, BOUNDS, SCALE), oracle.OracleFrame(R["kr"], R["dr"], None, BOUNDS, SCALE)
    dfl, dfr = msorb_mod.Frame(R["kl"], R["dl"], None, BOUNDS, SCALE), msorb_mod.Frame(R["kr"], R["dr"], None, BOUNDS, SCALE)
    try:
        for far, ratio in ((False, 0.8), (True, 0.6)):
            want_mp, got_mp = R["frame_mp"].copy(), R["frame_mp"].copy()
            wn = oracle.search_by_projection_mps_rig(ofl, ofr, R["mp"], R["l2r"], R["r2l"], want_mp, R["th"], far, 40.0, ratio)
            gn = msorb_mod.search_by_projection_mps_rig(dfl, dfr, R["mp"], R["l2r"], R["r2l"], got_mp, R["th"], far, 40.0, ratio)
            assert gn == wn and np.array_equal(got_mp, want_mp), (case, far, ratio, gn, wn, int((got_mp != want_mp).sum()))
            if case["M"] >= 2500 and case["n_left"] > 100 and case["n_right"] > 100:
                nl = case["n_left"]
                changed = got_mp != R["frame_mp"]
                assert changed[:nl].sum() > 100 and changed[nl:].sum() > 100          # both cameras matched
                # the coupling is exercised: some map point sits on both sides of a stereo pair
                both = [j for j in np.flatnonzero(R["l2r"] >= 0) if got_mp[j] >= 0 and got_mp[j] == got_mp[nl + R["l2r"][j]]]
                assert len(both) > 30
    finally:
        dfl.close(); dfr.close()


def test_two_camera_search_is_not_two_single_camera_searches(msorb_mod, oracle):
    """the passes are coupled: running the left and the right camera as two independent rectified-style searches gives another result"""
    R = make_rig(oracle, seed=11, n_left=1500, n_right=1400, M=4096)
    ofl, ofr = oracle.OracleFrame(R["kl"], R["dl"], None, BOUNDS, SCALE), oracle.OracleFrame(R["kr"], R["dr"], None, BOUNDS, SCALE)
    want = R["frame_mp"].copy()
    oracle.search_by_projection_mps_rig(ofl, ofr, R["mp"], R["l2r"], R["r2l"], want, 1.0, False, 50.0, 0.8)
    nl = len(R["kl"])
    indep_l, indep_r = R["frame_mp"][:nl].copy(), R["frame_mp"][nl:].copy()
    mpl = {k: R["mp"][k] for k in ("track_in_view", "bad", "sparsified", "proj_x", "proj_y", "track_depth", "level", "view_cos", "desc", "obs")}
    mpl["proj_xr"] = R["mp"]["proj_x"]
    ofl.SearchByProjection_mps(mpl, indep_l, 1.0, False, 50.0, 0.8)
    mpr = dict(mpl, track_in_view=(R["mp"]["track_in_view_r"] & (R["mp"]["level_r"] >= 0)).astype(np.uint8), proj_x=R["mp"]["proj_xr"],
               proj_y=R["mp"]["proj_yr"], proj_xr=R["mp"]["proj_xr"], level=np.maximum(R["mp"]["level_r"], 0), view_cos=R["mp"]["view_cos_r"],
               sparsified=np.zeros_like(R["mp"]["sparsified"]))
    ofr.SearchByProjection_mps(mpr, indep_r, 1.0, False, 50.0, 0.8)
    assert not (np.array_equal(indep_l, want[:nl]) and np.array_equal(indep_r, want[nl:]))


def test_two_camera_search_argument_checks(msorb_mod, oracle):
    R = make_rig(oracle, seed=12, n_left=300, n_right=300, M=500)
    dfl, dfr = msorb_mod.Frame(R["kl"], R["dl"], None, BOUNDS, SCALE), msorb_mod.Frame(R["kr"], R["dr"], None, BOUNDS, SCALE)
    try:
        bad = R["l2r"].copy(); bad[5] = 300
        with pytest.raises(msorb_mod.MsorbError):
            msorb_mod.search_by_projection_mps_rig(dfl, dfr, R["mp"], bad, R["r2l"], R["frame_mp"].copy(), 1.0)
        fm = R["frame_mp"].copy(); fm[3] = 500
        with pytest.raises(msorb_mod.MsorbError):
            msorb_mod.search_by_projection_mps_rig(dfl, dfr, R["mp"], R["l2r"], R["r2l"], fm, 1.0)
        mp = dict(R["mp"], level=np.where(np.arange(500) == 7, 9, R["mp"]["level"]).astype(np.int32), track_in_view=np.ones(500, np.uint8),
                  bad=np.zeros(500, np.uint8))
        with pytest.raises(msorb_mod.MsorbError):
            msorb_mod.search_by_projection_mps_rig(dfl, dfr, mp, R["l2r"], R["r2l"], R["frame_mp"].copy(), 1.0)
    finally:
        dfl.close(); dfr.close()


def gate_of(kl, kr):
    """pKF->GetKeyPoint(idx) for a right-camera index idx (KeyFrame.h:377-385): mvKeys[idx] below NLeft, mvKeysRight[idx - NLeft] beyond"""
    nl, nr = len(kl), len(kr)
    gate = np.zeros(nr, kl.dtype)
    j = np.arange(nr)
    gate[j < nl] = kl[j[j < nl]]
    gate[j >= nl] = kr[j[j >= nl] - nl]
    return gate


@pytest.mark.parametrize("seed,n_left,n_right,th", [(1, 1500, 1400, 3.0), (2, 600, 1500, 4.0), (3, 1500, 1400, 2.5), (4, 0, 700, 3.0)])
def test_fuse_search_right_camera_of_a_two_camera_keyframe(msorb_mod, oracle, seed, n_left, n_right, th):
    """msorb_fuse_search_gated = ORBmatcher::Fuse(pKF, vpMapPoints, th, bRight = true) (ORBmatcher.cc:1499-1561): the window walks the right
    camera's grid, the level band and the reprojection-error gate read GetKeyPoint(idx) / GetuRight(idx) with that right-camera index —
    a LEFT keypoint for idx < NLeft, mvKeysRight[idx - NLeft] beyond (n_left < n_right in case 2 reaches that arm)."""
    R = make_rig(oracle, seed=40 + seed, n_left=n_left, n_right=n_right, M=0)
    rng = np.random.Generator(np.random.PCG64(400 + seed))
    kl, kr, dr = R["kl"], R["kr"], R["dr"]
    if seed != 3 and n_left:  # the gate keypoint near the window keypoint (else the error gate rejects nearly everything): left keypoint j ~ right keypoint j
        m = min(n_left, n_right)
        kl["x"][:m] = kr["x"][:m] + rng.normal(0, 1.0, m); kl["y"][:m] = kr["y"][:m] + rng.normal(0, 1.0, m)
        kl["octave"][:m] = np.clip(kr["octave"][:m] + rng.integers(-1, 2, m), 0, 7)
    gate = gate_of(kl, kr)
    # the reference's KeyFrame has mvuRight = -1 for such a rig (Frame.cc:1069); the gate array is general: a third with a stereo term
    gur = np.where(rng.random(n_right) < (0.0 if seed == 1 else 0.3), gate["x"] - rng.uniform(1, 30, n_right), -1).astype(np.float32)
    inv_sigma2 = (np.float32(1.0) / (SCALE * SCALE)).astype(np.float32)
    M = 3000
    src = rng.integers(0, n_right, M)
    sig = rng.choice([0.3, 1.0, 3.0], M)
    u = (kr["x"][src] + rng.normal(0, sig)).astype(np.float32)
    v = (kr["y"][src] + rng.normal(0, sig)).astype(np.float32)
    ur = np.where(gur[src] >= 0, gur[src] + rng.normal(0, sig), u - 20).astype(np.float32)
    level = np.clip(gate["octave"][src] + rng.integers(0, 2, M), 0, 7).astype(np.int32)
    radius = (np.float32(th) * SCALE[level]).astype(np.float32)
    valid = (rng.random(M) < 0.9).astype(np.uint8)
    u[rng.random(M) < 0.02] = -50.0
    desc = mc.flip_bits(rng, dr[src], 40)
    F, O = msorb_mod.Frame(kr, dr, None, BOUNDS, SCALE), oracle.OracleFrame(kr, dr, None, BOUNDS, SCALE)
    try:
        bi, bd = F.FuseSearchGated(gate, None if seed == 1 else gur, inv_sigma2, valid, u, v, ur, level, radius, desc)
        wi, wd = O.FuseSearchGated(gate, None if seed == 1 else gur, inv_sigma2, valid, u, v, ur, level, radius, desc)
        assert np.array_equal(bi, wi) and np.array_equal(bd, wd)
        assert np.all(bi[valid == 0] == -1)
        # the gate array is what is read: gating on the right camera's own keypoints (= msorb_fuse_search) answers differently ...
        oi, od = F.FuseSearch(inv_sigma2, valid, u, v, ur, level, radius, desc)
        si, sd = F.FuseSearchGated(kr, None, inv_sigma2, valid, u, v, ur, level, radius, desc)
        assert np.array_equal(oi, si) and np.array_equal(od, sd)          # ... and with the frame's own keypoints it IS msorb_fuse_search
        if n_left:
            assert not np.array_equal(wi, oi)
            assert seed in (3,) or (wi >= 0).sum() > 100                      # (case 3: unrelated gate keypoints, the error gate rejects nearly all)
        else:
            # NLeft = 0: GetKeyPoint(idx) = mvKeysRight[idx] — the plain search on a frame that carries the gate's mvuRight
            assert np.array_equal(wi, oracle.OracleFrame(kr, dr, gur, BOUNDS, SCALE).FuseSearch(inv_sigma2, valid, u, v, ur, level, radius, desc)[0])
        if n_right > 3:
            with pytest.raises(msorb_mod.MsorbError):
                F.FuseSearchGated(_bad_octave(gate), None, inv_sigma2, valid, u, v, ur, level, radius, desc)   # octave outside mvInvLevelSigma2
    finally:
        F.close()


def _bad_octave(gate):
    g = gate.copy()
    g["octave"][3] = 9
    return g


# ---- through the drop-in CLASS (tests/dropin_rig_main.cc): Frame objects with Nleft != -1 ---------------------------------------
import os
import struct
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CAM = (718.856, 718.856, 607.19, 185.2)


def _scene(oracle, seed, motion):
    """A two-camera current frame whose right camera really is the left one moved by Trl (so both arms of SearchByProjection(Current,
    Last) find their points), a two-camera last frame with map points in the world, the local map of make_rig on top."""
    rng = np.random.Generator(np.random.PCG64(seed))
    fx, fy, cx, cy = CAM
    n_left = 1300
    kl = np.zeros(n_left, oracle.KP_DTYPE)
    kl["x"] = rng.uniform(30, 1210, n_left); kl["y"] = rng.uniform(30, 346, n_left)
    kl["octave"] = rng.integers(0, 8, n_left); kl["angle"] = rng.uniform(0, 360, n_left); kl["size"] = 31
    dl = rng.integers(0, 256, (n_left, 32), dtype=np.uint8)
    z = rng.uniform(4, 40, n_left).astype(np.float32)
    trl = np.array([-0.12, 0.0, 0.0], np.float32)
    Xc = np.stack([(kl["x"] - cx) * z / fx, (kl["y"] - cy) * z / fy, z], 1).astype(np.float32)
    ur = fx * (Xc[:, 0] + trl[0]) / Xc[:, 2] + cx
    part = np.flatnonzero((rng.random(n_left) < 0.6) & (ur > 25) & (ur < 1215))
    n_extra = 300
    n_right = len(part) + n_extra
    kr = np.zeros(n_right, oracle.KP_DTYPE)
    kr["x"][:len(part)] = ur[part] + rng.normal(0, 0.7, len(part)); kr["y"][:len(part)] = kl["y"][part] + rng.normal(0, 0.7, len(part))
    kr["octave"][:len(part)] = kl["octave"][part]; kr["angle"][:len(part)] = (kl["angle"][part] + rng.normal(0, 5, len(part))) % 360
    kr["x"][len(part):] = rng.uniform(30, 1210, n_extra); kr["y"][len(part):] = rng.uniform(30, 346, n_extra)
    kr["octave"][len(part):] = rng.integers(0, 8, n_extra); kr["angle"][len(part):] = rng.uniform(0, 360, n_extra); kr["size"] = 31
    dr = np.concatenate([mc.flip_bits(rng, dl[part], 20), rng.integers(0, 256, (n_extra, 32), dtype=np.uint8)])
    perm = rng.permutation(n_right)          # (partners must not sit at the front of the right camera's arrays)
    kr, dr = kr[perm], dr[perm]
    inv = np.empty(n_right, np.int64); inv[perm] = np.arange(n_right)
    l2r = np.full(n_left, -1, np.int32); r2l = np.full(n_right, -1, np.int32)
    l2r[part] = inv[np.arange(len(part))]; r2l[l2r[part]] = part
    # poses: current camera = identity rotation; the last frame a step behind / ahead / beside it
    tcw = np.array([0.3, -0.1, 0.2], np.float32)
    tlw = tcw + np.array({"forward": [0, 0, 1.5], "backward": [0, 0, -1.5], "side": [0.2, 0, 0.1]}[motion], np.float32)
    Xw = Xc - tcw
    n_last = 1500
    src = rng.integers(0, n_left, n_last)
    on_kp = rng.random(n_last) < 0.75
    pos = np.where(on_kp[:, None], Xw[src] + rng.normal(0, 0.01, (n_last, 3)), rng.uniform([-8, -3, 3], [8, 3, 50], (n_last, 3))).astype(np.float32)
    ll = 900
    lk = np.zeros(n_last, oracle.KP_DTYPE)
    lk["octave"] = np.clip(kl["octave"][src] + rng.integers(-1, 2, n_last), 0, 7); lk["angle"] = (kl["angle"][src] + rng.normal(0, 8, n_last)) % 360
    lk["x"] = rng.uniform(30, 1210, n_last); lk["y"] = rng.uniform(30, 346, n_last); lk["size"] = 31
    last = dict(kl=lk[:ll], kr=lk[ll:], has=(rng.random(n_last) < 0.85).astype(np.uint8), outlier=(rng.random(n_last) < 0.05).astype(np.uint8),
                pos=pos, desc=np.where(on_kp[:, None], mc.flip_bits(rng, dl[src], 30), rng.integers(0, 256, (n_last, 32), dtype=np.uint8)).astype(np.uint8),
                obs=np.where(rng.random(n_last) < 0.2, 0, rng.integers(1, 9, n_last)).astype(np.int32))
    eye = np.eye(3, dtype=np.float32).reshape(9)
    pose = np.concatenate([eye, tcw, eye, tlw, eye, trl]).astype(np.float32)
    cur_hold = np.where(rng.random(n_left + n_right) < 0.12, rng.integers(0, 4, n_left + n_right), -1).astype(np.int32)
    return dict(kl=kl, dl=dl, kr=kr, dr=dr, l2r=l2r, r2l=r2l, last=last, pose=pose, cur_hold=cur_hold, ll=ll)


@pytest.mark.parametrize("seed,motion,check_ori", [(21, "forward", True), (22, "backward", True), (23, "side", True), (24, "side", False)])
def test_class_search_by_projection_on_two_camera_frames(msorb_mod, oracle, tmp_path, seed, motion, check_ori):
    exe = str(tmp_path / "dropin_rig")
    subprocess.check_call(["g++", "-O2", "-std=c++17", f"-I{ROOT}/tests/slam_stub", f"-I{ROOT}/tests/cv_stub", f"-I{ROOT}/ms-slam_amd/host",
                           f"-I{ROOT}/include", f"{ROOT}/tests/dropin_rig_main.cc", f"{ROOT}/ms-slam_amd/host/ORBmatcher.cc", f"-L{ROOT}/ms-slam_amd",
                           "-lmsorb", f"-Wl,-rpath,{ROOT}/ms-slam_amd", "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib", "-lpthread", "-o", exe])
    S = _scene(oracle, seed, motion)
    nl, nr = len(S["kl"]), len(S["kr"])
    N = nl + nr
    M = 3500
    R = make_rig(oracle, seed + 100, nl, nr, M)          # only its map-point table and frame_mp are used, re-aimed at THIS frame's keypoints
    rng = np.random.Generator(np.random.PCG64(seed + 200))
    sl, sr = rng.integers(0, nl, M), rng.integers(0, nr, M)
    mp = R["mp"]
    mp["proj_x"] = (S["kl"]["x"][sl] + rng.normal(0, 2.5, M)).astype(np.float32); mp["proj_y"] = (S["kl"]["y"][sl] + rng.normal(0, 2.5, M)).astype(np.float32)
    has_p = S["l2r"][sl] >= 0
    sr = np.where(has_p, S["l2r"][sl], sr)
    mp["proj_xr"] = (S["kr"]["x"][sr] + rng.normal(0, 2.5, M)).astype(np.float32); mp["proj_yr"] = (S["kr"]["y"][sr] + rng.normal(0, 2.5, M)).astype(np.float32)
    mp["level"] = np.clip(S["kl"]["octave"][sl] + rng.integers(0, 2, M), 0, 7).astype(np.int32)
    mp["level_r"] = np.where(rng.random(M) < 0.05, -1, np.clip(S["kr"]["octave"][sr] + rng.integers(0, 2, M), 0, 7)).astype(np.int32)
    mp["desc"] = np.where((rng.random(M) < 0.85)[:, None], mc.flip_bits(rng, S["dl"][sl], 45), rng.integers(0, 256, (M, 32), dtype=np.uint8)).astype(np.uint8)
    frame_mp0 = R["frame_mp"]
    th13, th_far, ratio, far, th14, mb, mono = 2.0, 45.0, 0.8, 1, 7.0, 0.537, 0
    with open(tmp_path / "in.bin", "wb") as f:
        f.write(struct.pack("<6i", nl, nr, M, S["ll"], len(S["last"]["has"]) - S["ll"], 8))
        f.write(struct.pack("<16f", *CAM, *BOUNDS, th13, th_far, ratio, far, th14, mb, mono, float(check_ori)))
        f.write(SCALE.tobytes())
        for a in (S["kl"], S["kr"], np.concatenate([S["dl"], S["dr"]]), S["l2r"], S["r2l"]):
            f.write(np.ascontiguousarray(a).tobytes())
        for k, dt in (("track_in_view", np.uint8), ("track_in_view_r", np.uint8), ("bad", np.uint8), ("sparsified", np.uint8), ("proj_x", np.float32),
                      ("proj_y", np.float32), ("proj_xr", np.float32), ("proj_yr", np.float32), ("track_depth", np.float32), ("level", np.int32),
                      ("level_r", np.int32), ("view_cos", np.float32), ("view_cos_r", np.float32), ("desc", np.uint8), ("obs", np.int32)):
            f.write(np.ascontiguousarray(mp[k], dt).tobytes())
        f.write(frame_mp0.tobytes())
        L = S["last"]
        for a in (L["kl"], L["kr"], L["has"], L["outlier"], L["pos"], L["desc"], L["obs"], S["pose"], S["cur_hold"]):
            f.write(np.ascontiguousarray(a).tobytes())
        # tail: a KeyFrame for SearchByBoW(pKF, F) — its features are noisy copies of the frame's (both cameras), BoW nodes by source
        tail = f.tell()
        NK = 1100
        allk = np.concatenate([S["kl"], S["kr"]]); alld = np.concatenate([S["dl"], S["dr"]])
        fnode = (np.arange(N) * 7919 % 60).astype(np.int32)
        part = np.flatnonzero(S["l2r"] >= 0)
        fnode[nl + S["l2r"][part]] = fnode[part]                       # stereo partners share their node
        ksrc = rng.integers(0, nl, NK)
        kk = np.zeros(NK, oracle.KP_DTYPE); kk["angle"] = (allk["angle"][ksrc] + 30 + rng.normal(0, 4, NK)) % 360; kk["octave"] = allk["octave"][ksrc]
        kd = mc.flip_bits(rng, alld[ksrc], 20)
        knode = np.where(rng.random(NK) < 0.05, -1, fnode[ksrc]).astype(np.int32)
        kstate = rng.choice(np.array([0, 1, 1, 1, 1, 2], np.uint8), NK)
        f.write(struct.pack("<i", NK))
        for a in (kk, kd, knode, kstate, fnode):
            f.write(np.ascontiguousarray(a).tobytes())
        f.write(struct.pack("<q", tail))
    p = subprocess.run([exe, str(tmp_path / "in.bin"), str(tmp_path / "out.bin")], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    blob = (tmp_path / "out.bin").read_bytes()
    pos = 0

    def take(dt, n):
        nonlocal pos
        a = np.frombuffer(blob, dt, n, pos)
        pos += a.nbytes
        return a
    # ---- SearchByProjection(F, vpMapPoints, th, bFarPoints, thFarPoints) through the class == the oracle's two-camera arm
    nm13 = int(take(np.int32, 1)[0]); got13 = take(np.int32, N); refused = int(take(np.int32, 1)[0])
    ofl, ofr = oracle.OracleFrame(S["kl"], S["dl"], None, BOUNDS, SCALE), oracle.OracleFrame(S["kr"], S["dr"], None, BOUNDS, SCALE)
    want13 = frame_mp0.copy()
    wn13 = oracle.search_by_projection_mps_rig(ofl, ofr, mp, S["l2r"], S["r2l"], want13, th13, bool(far), th_far, ratio)
    assert nm13 == wn13 and np.array_equal(got13, want13)
    assert (want13[:nl] != frame_mp0[:nl]).sum() > 200 and (want13[nl:] != frame_mp0[nl:]).sum() > 100 and refused == 1
    # ---- SearchByProjection(CurrentFrame, LastFrame, th, bMono) through the class == the oracle's arm on the class's own projections
    nm14 = int(take(np.int32, 1)[0]); got14 = take(np.int32, N)
    fwd, bwd = (int(x) for x in take(np.int32, 2))
    n_last = len(S["last"]["has"])
    last = dict(valid=take(np.uint8, n_last), u=take(np.float32, n_last), v=take(np.float32, n_last), u_r=take(np.float32, n_last),
                v_r=take(np.float32, n_last), octave=take(np.int32, n_last), angle=take(np.float32, n_last), desc=S["last"]["desc"],
                mp=np.arange(n_last, dtype=np.int32))
    nm_bow = int(take(np.int32, 1)[0]); got_bow = take(np.int32, N)
    assert pos == len(blob)
    pb = dict(desc1=kd, desc2=alld, valid1=(kstate == 1).astype(np.uint8), fv1=bmc.feature_vector_from_nodes(knode),
              fv2=bmc.feature_vector_from_nodes(fnode), angle1=kk["angle"], angle2=allk["angle"])
    wn_bow, w21 = oracle.search_by_bow_rig(pb, nl, 50, 0.7, check_ori)
    assert nm_bow == wn_bow and np.array_equal(got_bow, w21), (nm_bow, wn_bow)
    assert (w21[:nl] >= 0).sum() > 100 and (w21[nl:] >= 0).sum() > 50
    assert (fwd, bwd) == {"forward": (1, 0), "backward": (0, 1), "side": (0, 0)}[motion]
    assert last["valid"].sum() > 600 and np.array_equal(last["octave"][last["valid"] > 0],
                                                        np.concatenate([S["last"]["kl"], S["last"]["kr"]])["octave"][last["valid"] > 0])
    held = np.flatnonzero(S["cur_hold"] >= 0)
    cur = np.full(N, -1, np.int32)
    cur[held] = n_last + np.arange(len(held))
    last["obs"] = np.concatenate([S["last"]["obs"], S["cur_hold"][held]]).astype(np.int32)
    want14 = cur.copy()
    wn14 = oracle.search_by_projection_frames_rig(ofl, ofr, last, want14, th14, bool(fwd), bool(bwd), check_ori)
    want_ids = np.where(want14 >= n_last, -2, want14)
    assert nm14 == wn14 and np.array_equal(got14, want_ids), (nm14, wn14, int((got14 != want_ids).sum()))
    newl, newr = ((want14[:nl] >= 0) & (want14[:nl] < n_last)).sum(), ((want14[nl:] >= 0) & (want14[nl:] < n_last)).sum()
    assert newl > 150 and newr > 60, (newl, newr)                      # both arms matched
    # and the flat C entry on the same projections gives the same (the class adds nothing to it but the projection)
    dfl, dfr = msorb_mod.Frame(S["kl"], S["dl"], None, BOUNDS, SCALE), msorb_mod.Frame(S["kr"], S["dr"], None, BOUNDS, SCALE)
    try:
        c2 = cur.copy()
        assert msorb_mod.search_by_projection_frames_rig(dfl, dfr, last, c2, th14, bool(fwd), bool(bwd), check_ori) == wn14 and np.array_equal(c2, want14)
    finally:
        dfl.close(); dfr.close()


# ---- SearchByBoW(pKF, F, vpMapPointMatches) on a two-camera frame (ORBmatcher.cc:223-421 with the Nleft arms) -----------------
import bow_match_cases as bmc


def _bow_rig_pair(seed, n1, n2, n_left, n_nodes=50, shuffle=False):
    p = bmc.make_pair(seed, n1=n1, n2=n2, n_nodes=n_nodes, shuffle_lists=shuffle)
    rng = np.random.default_rng(seed + 5000)
    node2 = np.full(n2, -1, np.int64)
    nodes, begin, feat = p["fv2"]
    for r in range(len(nodes)):
        node2[feat[begin[r]:begin[r + 1]]] = nodes[r]
    # right-camera rows that are the stereo partners of left rows: the same node, a near descriptor
    if 0 < n_left < n2:
        right = np.arange(n_left, n2)
        twin = right[rng.random(len(right)) < 0.6]
        src = rng.integers(0, n_left, len(twin))
        p["desc2"][twin] = bmc.bow_cases._flip_bits(rng, p["desc2"][src], rng.integers(0, 14, len(twin)))
        node2[twin] = node2[src]
        p["angle2"][twin] = np.mod(p["angle2"][src] + rng.normal(0, 3, len(twin)), 360).astype(np.float32)
    p["fv2"] = bmc.feature_vector_from_nodes(node2)
    if shuffle:
        for r in range(len(p["fv2"][0])):
            rng.shuffle(p["fv2"][2][p["fv2"][1][r]:p["fv2"][1][r + 1]])
    p["avail2"] = None
    return p


@pytest.mark.parametrize("seed,n1,n2,n_left,shuffle", [(1, 1200, 2400, 1300, False), (2, 900, 1800, 900, True), (3, 1500, 1000, 1000, False),
                                                      (4, 700, 1400, 0, False), (5, 800, 1600, 30, True), (6, 0, 500, 250, False),
                                                      (7, 600, 0, 0, False), (8, 2000, 3000, 1500, True)])
def test_search_by_bow_two_camera_frame(msorb_mod, oracle, seed, n1, n2, n_left, shuffle):
    p = _bow_rig_pair(seed, n1, n2, n_left, shuffle=shuffle)
    for ratio, ori in ((0.7, True), (0.9, False), (0.6, True)):
        wn, w21 = oracle.search_by_bow_rig(p, n_left, 50, ratio, ori)
        gn, g21, g12 = msorb_mod.search_by_bow_rig(p, n_left, 50, ratio, ori)
        assert gn == wn and np.array_equal(g21, w21), (seed, ratio, ori, gn, wn, int((g21 != w21).sum()))
        for k in np.flatnonzero(g12 >= 0):                       # match12 = the left partner
            assert g12[k] < n_left and g21[g12[k]] == k
        if n1 >= 900 and n_left >= 900 and n2 - n_left >= 900:
            assert (w21[:n_left] >= 0).sum() > 100 and (w21[n_left:] >= 0).sum() > 50      # both cameras matched
            both = set(w21[:n_left][w21[:n_left] >= 0]) & set(w21[n_left:][w21[n_left:] >= 0])
            assert len(both) > 30                                # KeyFrame features matched in BOTH cameras
    if n1 >= 900 and n_left >= 900 and n2 - n_left >= 900:
        # the arm is not two one-camera searches: a right match needs the LEFT best distance <= TH_LOW and passes without a ratio test
        _, _, m21_all = oracle.search_by_bow(p["desc1"], p["desc2"], p["valid1"], None, p["fv1"], p["fv2"], p["angle1"], p["angle2"], 50, True, 0.7, True)
        assert not np.array_equal(m21_all, oracle.search_by_bow_rig(p, n_left, 50, 0.7, True)[1])


# ---- SearchForTriangulation between KeyFrames of a two-camera rig: the geometric test stays with the caller -----------------------
def _accept_fn(seed, n1, n2, rate):
    """a stand-in for pCamera1->epipolarConstrain(pCamera2, kp1, kp2, R12, t12, ...) (ORBmatcher.cc:1332): a pure, pseudo-random predicate
    of the two feature indices"""
    table = np.random.default_rng(seed).random((max(n1, 1), max(n2, 1))) < rate
    return lambda i1, i2: bool(table[i1, i2])


@pytest.mark.parametrize("seed,n1,n2,rate,shuffle", [(1, 1200, 1300, 0.5, False), (2, 900, 2000, 0.2, True), (3, 1500, 1000, 0.9, False),
                                                     (4, 0, 500, 0.5, False), (5, 600, 0, 0.5, False), (6, 2500, 2500, 0.5, True),
                                                     (7, 800, 900, 0.0, False), (8, 800, 900, 1.0, True)])
def test_search_for_triangulation_with_the_callers_geometric_test(msorb_mod, oracle, seed, n1, n2, rate, shuffle):
    """msorb_search_for_triangulation_cb vs the oracle's statement-by-statement arm (orc_search_for_triangulation_rig): same matches
    although the library asks accept() in another order (best distance first) than the reference's running-minimum scan."""
    import bow_match_cases as bmc
    p = bmc.make_pair(300 + seed, n1, n2, n_nodes=25, flip=30, dup_frac=0.25, shuffle_lists=shuffle)
    acc = _accept_fn(seed, n1, n2, rate)
    for ori in (True, False):
        wn, w12, wcalls = oracle.search_for_triangulation_rig(p, acc, False, ori)
        gn, g12, calls = msorb_mod.search_for_triangulation_cb(p, acc, 50, ori)
        assert gn == wn and np.array_equal(g12, w12), (seed, ori, gn, wn, int((g12 != w12).sum()))
        if n1 >= 800 and n2 >= 900:
            assert (wn > 30) is (rate > 0) and len(calls) > 0
            d1, d2 = p["desc1"], p["desc2"]
            for i1, i2 in calls[:200]:                             # accept is only asked about eligible pairs within TH_LOW
                assert p["valid1"][i1] and p["avail2"][i2] and int(np.unpackbits(d1[i1] ^ d2[i2]).sum()) <= 50
    # bCoarse (:1332 `bCoarse || ...`): every candidate passes
    wn, w12, _ = oracle.search_for_triangulation_rig(p, acc, True, True)
    gn, g12, _ = msorb_mod.search_for_triangulation_cb(p, lambda i1, i2: True, 50, True)
    assert gn == wn and np.array_equal(g12, w12)
    # all trains eligible (avail2 = NULL)
    q = dict(p, avail2=None)
    wn, w12, _ = oracle.search_for_triangulation_rig(q, acc, False, True)
    gn, g12, _ = msorb_mod.search_for_triangulation_cb(q, acc, 50, True)
    assert gn == wn and np.array_equal(g12, w12)


def test_the_accept_order_matters_only_through_purity(msorb_mod, oracle):
    """ties in distance: the LAST train of equal distance wins in the reference (`dist > bestDist` lets an equal one replace, :1277); an accept
    that rejects that one hands the match to the earlier one"""
    import bow_match_cases as bmc
    p = bmc.make_pair(77, 600, 700, n_nodes=6, flip=0, dup_frac=1.0, mask_frac=0.0, far_frac=0.0)   # exact duplicates: many ties at distance 0
    every = lambda i1, i2: True
    wn, w12, _ = oracle.search_for_triangulation_rig(p, every, False, False)
    gn, g12, _ = msorb_mod.search_for_triangulation_cb(p, every, 50, False)
    assert gn == wn and np.array_equal(g12, w12) and wn > 300
    odd = lambda i1, i2: (i1 + i2) % 2 == 1
    wn, w12, _ = oracle.search_for_triangulation_rig(p, odd, False, False)
    gn, g12, _ = msorb_mod.search_for_triangulation_cb(p, odd, 50, False)
    assert gn == wn and np.array_equal(g12, w12)
    with pytest.raises(msorb_mod.MsorbError):
        bad = dict(p, fv1=(p["fv1"][0][::-1].copy(), p["fv1"][1], p["fv1"][2]))
        msorb_mod.search_for_triangulation_cb(bad, every, 50, False)


# ---- KeyFrames of a two-camera rig through the drop-in CLASS (tests/dropin_rig_kf_main.cc) ---------------------------------------
def _kf_cam(rng, oracle, n):
    k = np.zeros(n, oracle.KP_DTYPE)
    xs = rng.permutation(np.arange(40, 40 + 4 * n))[:n] * 0.25 + 0.125        # distinct coordinates: the epipolar log is keyed on them
    k["x"] = 20 + xs % 1190; k["y"] = 20 + rng.uniform(0, 330, n)
    k["octave"] = rng.integers(0, 8, n); k["angle"] = rng.uniform(0, 360, n); k["size"] = 31
    return k


@pytest.mark.parametrize("seed,check_ori", [(31, True), (32, False)])
def test_class_triangulation_and_fuse_on_two_camera_keyframes(msorb_mod, oracle, tmp_path, seed, check_ori):
    """ORBmatcher::SearchForTriangulation between two KeyFrames of a two-camera rig (ORBmatcher.cc:1168-1402: the camera model's
    epipolarConstrain called back with the cameras / relative pose of :1294-1330) and ORBmatcher::Fuse(pKF, vpMapPoints, th, bRight) for
    both cameras (:1404-1597), against the oracle's arms."""
    exe = str(tmp_path / "dropin_rig_kf")
    subprocess.check_call(["g++", "-O2", "-std=c++17", f"-I{ROOT}/tests/slam_stub", f"-I{ROOT}/tests/cv_stub", f"-I{ROOT}/ms-slam_amd/host",
                           f"-I{ROOT}/include", f"{ROOT}/tests/dropin_rig_kf_main.cc", f"{ROOT}/ms-slam_amd/host/ORBmatcher.cc", f"-L{ROOT}/ms-slam_amd",
                           "-lmsorb", f"-Wl,-rpath,{ROOT}/ms-slam_amd", "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib", "-lpthread", "-o", exe])
    rng = np.random.Generator(np.random.PCG64(seed))
    NL1, NR1, NL2, NR2, M = 900, 800, 850, 900, 2500
    N1, N2 = NL1 + NR1, NL2 + NR2
    cam0 = np.array(CAM, np.float32); cam1 = np.array([CAM[0] * 1.01, CAM[1] * 0.99, CAM[2] + 3.0, CAM[3] - 2.0], np.float32)
    sigma2 = (SCALE * SCALE).astype(np.float32)
    # KeyFrame 2: random features; KeyFrame 1: noisy copies of KeyFrame 2's (either camera of either KeyFrame: all four side pairs occur)
    k2l, k2r = _kf_cam(rng, oracle, NL2), _kf_cam(rng, oracle, NR2)
    k2 = np.concatenate([k2l, k2r])
    d2 = rng.integers(0, 256, (N2, 32), dtype=np.uint8)
    node2 = (rng.integers(0, 40, N2) * 3 + 5).astype(np.int32)
    twin = rng.integers(0, N2, N2 // 4)                               # near-duplicates: several trains within TH_LOW of one query
    d2[twin] = mc.flip_bits(rng, d2[(twin + 1) % N2], 8); node2[twin] = node2[(twin + 1) % N2]
    node2[rng.random(N2) < 0.03] = -1
    k1l, k1r = _kf_cam(rng, oracle, NL1), _kf_cam(rng, oracle, NR1)
    # the right-camera Fuse pass gates on GetKeyPoint(idx) with a right-camera index: the LEFT keypoint of the same number (KeyFrame.h:377-385).
    # Left keypoint j near right keypoint j for a share of them, so that this gate passes often enough to be seen
    m = min(NL1, NR1)
    near = rng.random(m) < 0.6
    k1l["x"][:m][near] = k1r["x"][:m][near] + rng.normal(0, 0.8, int(near.sum())); k1l["y"][:m][near] = k1r["y"][:m][near] + rng.normal(0, 0.8, int(near.sum()))
    k1l["octave"][:m][near] = k1r["octave"][:m][near]
    k1 = np.concatenate([k1l, k1r])
    src = rng.integers(0, N2, N1)
    d1 = mc.flip_bits(rng, d2[src], 25)
    node1 = np.where(rng.random(N1) < 0.08, rng.integers(0, 40, N1) * 3 + 6, node2[src]).astype(np.int32)
    k1["angle"] = (k2["angle"][src] + 40 + rng.normal(0, 5, N1)) % 360
    k1l, k1r = k1[:NL1], k1[NL1:]
    held1 = (rng.random(N1) < 0.3).astype(np.uint8); held2 = (rng.random(N2) < 0.3).astype(np.uint8)
    obs1 = rng.integers(1, 6, N1).astype(np.int32); obs2 = rng.integers(1, 6, N2).astype(np.int32)
    eye = np.eye(3)
    t1 = np.array([0.3, -0.2, 0.1]); t2 = np.array([-0.4, 0.15, 0.25]); trl = np.array([-0.12, 0.013, 0.004])
    c, s_ = np.cos(0.05), np.sin(0.05)
    R2 = np.array([[c, -s_, 0], [s_, c, 0], [0, 0, 1]])

    def pose_bytes(R, t):
        return np.concatenate([R.reshape(9), t, eye.reshape(9), trl]).astype(np.float32)
    # ---- Fuse candidates for KeyFrame 1 (Tcw = [I | t1]): points in front of either camera that project near one of its keypoints
    side = rng.random(M) < 0.5                                        # True: aimed at the right camera
    il, ir = rng.integers(0, NL1, M), rng.integers(0, NR1, M)         # the keypoint aimed at, per camera
    gate1 = gate_of(k1l, k1r)
    px = np.where(side, k1r["x"][ir], k1l["x"][il]) + rng.normal(0, 0.7, M)
    py = np.where(side, k1r["y"][ir], k1l["y"][il]) + rng.normal(0, 0.7, M)
    cm = np.where(side[:, None], cam1[None, :], cam0[None, :]).astype(np.float64)
    z = rng.uniform(5, 40, M)
    Xcam = np.stack([(px - cm[:, 2]) / cm[:, 0] * z, (py - cm[:, 3]) / cm[:, 1] * z, z], 1)
    Xcam[rng.random(M) < 0.03, 2] *= -1
    Xw = np.where(side[:, None], Xcam - trl - t1, Xcam - t1).astype(np.float32)
    Ow = np.where(side[:, None], -(t1 + trl), -t1)
    PO = Xw - Ow
    dist = np.linalg.norm(PO, axis=1)
    normal = PO / dist[:, None] + rng.normal(0, 0.3, (M, 3))
    normal = (normal / np.linalg.norm(normal, axis=1, keepdims=True)).astype(np.float32)
    lvl = np.where(side, gate1["octave"][ir], k1l["octave"][il])      # the level the gate will read (right pass: the gate keypoint's)
    maxd = (dist * SCALE[lvl] * rng.uniform(0.95, 1.05, M)).astype(np.float32)
    mind = (maxd / SCALE[7] * rng.uniform(0.5, 1.2, M)).astype(np.float32)
    state = rng.choice([0, 1, 2, 3], M, p=[0.03, 0.87, 0.05, 0.05]).astype(np.uint8)
    obs = rng.integers(0, 6, M).astype(np.int32)
    dsrc = np.where(side[:, None], d1[NL1:][ir], d1[:NL1][il])
    mdesc = mc.flip_bits(rng, dsrc, 30)
    th_fuse, mbf = 3.0, 386.1448
    with open(tmp_path / "in.bin", "wb") as f:
        f.write(struct.pack("<6i", NL1, NR1, NL2, NR2, M, 8))
        f.write(struct.pack("<16f", *cam0, *cam1, *BOUNDS, th_fuse, mbf, float(check_ori), 0.0))
        f.write(SCALE.tobytes()); f.write(sigma2.tobytes())
        for (kl_, kr_, d_, node_, held_, obs_, pose_) in ((k1l, k1r, d1, node1, held1, obs1, pose_bytes(eye, t1)), (k2l, k2r, d2, node2, held2, obs2, pose_bytes(R2, t2))):
            for a in (kl_, kr_, d_, node_, held_, obs_, pose_):
                f.write(np.ascontiguousarray(a).tobytes())
        for a in (state, Xw, normal, maxd, mind, obs, mdesc):
            f.write(np.ascontiguousarray(a).tobytes())
    p = subprocess.run([exe, str(tmp_path / "in.bin"), str(tmp_path / "out.bin")], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    blob = (tmp_path / "out.bin").read_bytes()
    pos = 0

    def take(dt, n):
        nonlocal pos
        a = np.frombuffer(blob, dt, n, pos)
        pos += a.nbytes
        return a
    # ---- SearchForTriangulation: the four relative poses of :1196-1200 in float64 (Tcw1 = [I | t1], Tcw2 = [R2 | t2], Trl = [I | trl])
    def se3(R, t): return (np.asarray(R, np.float64), np.asarray(t, np.float64))
    def mul(a, b): return (a[0] @ b[0], a[0] @ b[1] + a[1])
    def inv(a): return (a[0].T, -a[0].T @ a[1])
    T1w, T2w, Trl_ = se3(eye, t1), se3(R2, t2), se3(eye, trl)
    Tw2, Tr1w, Twr2 = inv(T2w), mul(Trl_, T1w), mul(inv(T2w), inv(Trl_))
    T12 = {(0, 0): mul(T1w, Tw2), (0, 1): mul(T1w, Twr2), (1, 0): mul(Tr1w, Tw2), (1, 1): mul(Tr1w, Twr2)}
    for T in T12.values():
        assert abs(T[1][0]) > 1e-3 and abs(T[1][1]) > 1e-3            # the stand-in predicate reads these signs

    def accept(i1, i2):
        r1, r2 = int(i1 >= NL1), int(i2 >= NL2)
        t12 = T12[(r1, r2)][1]
        h = (int(k1["x"][i1]) * 7 + int(k1["y"][i1]) * 13 + int(k2["x"][i2]) * 29 + int(k2["y"][i2]) * 31 + (1 + r1) * 5 + (1 + r2) * 11 +
             (3 if t12[0] > 0 else 0) + (17 if t12[1] > 0 else 0))
        return h % 3 != 0
    fv1, fv2 = bmc.feature_vector_from_nodes(node1), bmc.feature_vector_from_nodes(node2)
    at1 = {(float(k1["x"][i]), float(k1["y"][i])): i for i in range(N1)}
    at2 = {(float(k2["x"][i]), float(k2["y"][i])): i for i in range(N2)}
    assert len(at1) == N1 and len(at2) == N2
    for only_stereo, coarse in ((0, 0), (0, 1), (1, 0)):
        nm = int(take(np.int32, 1)[0]); npairs = int(take(np.int32, 1)[0]); pairs = take(np.int32, 2 * npairs).reshape(-1, 2)
        nrows = int(take(np.int32, 1)[0]); rows = take(np.float32, 21 * nrows).reshape(-1, 21)
        q = dict(desc1=d1, desc2=d2, valid1=((held1 == 0) & (not only_stereo)).astype(np.uint8), avail2=((held2 == 0) & (not only_stereo)).astype(np.uint8),
                 fv1=fv1, fv2=fv2, angle1=k1["angle"], angle2=k2["angle"])
        wn, w12, _ = oracle.search_for_triangulation_rig(q, accept, bool(coarse), check_ori)
        want = np.stack([np.flatnonzero(w12 >= 0), w12[w12 >= 0]], 1)
        assert nm == wn == npairs and np.array_equal(pairs, want), (only_stereo, coarse, nm, wn)
        if only_stereo:
            assert nm == 0 and nrows == 0                              # bStereo is false for every feature of such a KeyFrame (:1243-1247)
        elif coarse:
            assert nm > 200 and nrows == 0                             # `bCoarse || ...`: the camera model is never asked (:1332)
        else:
            assert nm > 150 and nrows >= nm
            sides = set()
            for r in rows:
                i1, i2 = at1[(float(r[0]), float(r[1]))], at2[(float(r[2]), float(r[3]))]
                r1, r2 = int(i1 >= NL1), int(i2 >= NL2)
                sides.add((r1, r2))
                assert (int(r[4]), int(r[5])) == (1 + r1, 1 + r2)                                        # mpCamera / mpCamera2 by the feature's image
                assert np.allclose(r[6:15], T12[(r1, r2)][0].reshape(9), atol=1e-5) and np.allclose(r[15:18], T12[(r1, r2)][1], atol=1e-5)
                assert r[18] == sigma2[k1["octave"][i1]] and r[19] == sigma2[k2["octave"][i2]]
                assert bool(r[20]) == accept(i1, i2)
            assert sides == {(0, 0), (0, 1), (1, 0), (1, 1)}
            assert (pairs[:, 0] >= NL1).sum() > 30 and (pairs[:, 1] >= NL2).sum() > 30
    assert int(take(np.int32, 1)[0]) == 2                              # two-camera against one-camera KeyFrame: refused both ways
    # ---- Fuse, left camera then right camera on the map the left pass left
    inv_sigma2 = (np.float32(1.0) / sigma2).astype(np.float32)
    ofl = oracle.OracleFrame(k1l, d1[:NL1], np.full(NL1, -1, np.float32), BOUNDS, SCALE)
    ofr = oracle.OracleFrame(k1r, d1[NL1:], None, BOUNDS, SCALE)
    gate, gur = gate1, np.full(NR1, -1, np.float32)
    pts = {i: dict(obs=int(obs[i]), bad=state[i] == 2, inkf=state[i] == 3) for i in range(M) if state[i]}
    kf_mp = {}
    for j in range(N1):
        if held1[j]:
            pts[100000 + j] = dict(obs=int(obs1[j]), bad=False, inkf=True)
            kf_mp[j] = 100000 + j
    for right in (0, 1):
        n_fused = int(take(np.int32, 1)[0])
        valid = take(np.uint8, M); u = take(np.float32, M); v = take(np.float32, M); ur = take(np.float32, M)
        level = take(np.int32, M); radius = take(np.float32, M)
        nlog = int(take(np.int32, 1)[0]); log = take(np.int64, 3 * nlog).reshape(-1, 3)
        alive = np.array([bool(state[i]) and not pts[i]["bad"] and not pts[i]["inkf"] for i in range(M)])
        assert np.all(valid[~alive] == 0) and valid.sum() > 600
        ok = (valid > 0) & (side == bool(right))                      # the geometry against float64, for the points aimed at this camera
        cmr = cam1 if right else cam0
        Xc = Xw.astype(np.float64) + t1 + (trl if right else 0)
        assert np.allclose(u[ok], cmr[0] * Xc[ok, 0] / Xc[ok, 2] + cmr[2], atol=3e-2) and np.allclose(v[ok], cmr[1] * Xc[ok, 1] / Xc[ok, 2] + cmr[3], atol=3e-2)
        assert ok.sum() > 400
        if right:
            bi, bd = ofr.FuseSearchGated(gate, gur, inv_sigma2, valid, u, v, ur, level, radius, mdesc)
        else:
            bi, bd = ofl.FuseSearch(inv_sigma2, valid, u, v, ur, level, radius, mdesc)
        off = NL1 if right else 0
        want_log, want_fused = [], 0
        for i in range(M):
            if not state[i] or pts[i]["bad"] or pts[i]["inkf"] or not valid[i]:
                continue
            if bd[i] <= 50:
                j = int(bi[i]) + off                                   # `if(bRight) idx += pKF->GetNLeft()` (:1547)
                if j in kf_mp:
                    x = kf_mp[j]
                    if not pts[x]["bad"]:
                        a, b = (i, x) if pts[x]["obs"] > pts[i]["obs"] else (x, i)       # a->Replace(b)
                        want_log.append((1, a, b))
                        pts[a]["bad"] = True
                        pts[b]["obs"] += pts[a]["obs"]
                        pts[b]["inkf"] = pts[b]["inkf"] or pts[a]["inkf"]
                else:
                    want_log.append((2, i, j))
                    pts[i]["inkf"] = True
                    pts[i]["obs"] += 2
                    kf_mp[j] = i
                want_fused += 1
        ids = lambda e: (e[0], e[1] - 5000000 if e[1] >= 5000000 else e[1], (e[2] - 5000000 if e[2] >= 5000000 else e[2]) if e[0] == 1 else e[2])
        got_log = [ids(tuple(int(x) for x in e)) for e in log]
        assert n_fused == want_fused and got_log == want_log, (right, n_fused, want_fused)
        assert n_fused > (60 if right else 250), (right, n_fused)
        if right:
            assert all(e[2] >= NL1 for e in want_log if e[0] == 2)     # right-camera observations land at idx + NLeft
    assert pos == len(blob)


# ---- SearchByProjection(CurrentFrame, LastFrame, th, bMono) on a two-camera CurrentFrame, at the C ABI on synthetic projection tables ----
def make_last_table(oracle, seed, n_left, n_right, n_last, th):
    """a last frame's projected map points aimed at the current frame's keypoints: several per keypoint (claims collide), some only
    seen by one camera, keypoints the frame already holds (with and without observations)"""
    R = make_rig(oracle, seed, n_left, n_right, 0, dense=seed % 2 == 0)
    rng = np.random.Generator(np.random.PCG64(seed + 5))
    kl, kr, dl, dr = R["kl"], R["kr"], R["dl"], R["dr"]
    il, ir = rng.integers(0, max(n_left, 1), n_last), rng.integers(0, max(n_right, 1), n_last)
    partner = (R["l2r"][il] >= 0) if n_left else np.zeros(n_last, bool)
    if n_left and n_right:
        ir = np.where(partner, R["l2r"][il], ir)
    noise = th * 0.4
    last = dict(valid=(rng.random(n_last) < 0.9).astype(np.uint8),
                u=((kl["x"][il] if n_left else np.full(n_last, -500.0)) + rng.normal(0, noise, n_last)).astype(np.float32),
                v=((kl["y"][il] if n_left else np.full(n_last, -500.0)) + rng.normal(0, noise, n_last)).astype(np.float32),
                u_r=((kr["x"][ir] if n_right else np.full(n_last, -500.0)) + rng.normal(0, noise, n_last)).astype(np.float32),
                v_r=((kr["y"][ir] if n_right else np.full(n_last, -500.0)) + rng.normal(0, noise, n_last)).astype(np.float32),
                octave=np.clip((kl["octave"][il] if n_left else kr["octave"][ir] if n_right else np.zeros(n_last, np.int32)) + rng.integers(-1, 2, n_last), 0, 7).astype(np.int32),
                angle=(((kl["angle"][il] if n_left else np.zeros(n_last)) + 20 + rng.normal(0, 6, n_last)) % 360).astype(np.float32),
                desc=(mc.flip_bits(rng, dl[il], 35) if n_left else rng.integers(0, 256, (n_last, 32), dtype=np.uint8)), mp=np.arange(n_last, dtype=np.int32))
    wide = rng.random(n_last) < 0.1
    last["u"][wide] += rng.normal(0, 6 * th, int(wide.sum())).astype(np.float32)          # left window empty for some: their right arm must not run (:2003-2004)
    N = n_left + n_right
    held = np.flatnonzero(rng.random(N) < 0.12)
    cur = np.full(N, -1, np.int32)
    cur[held] = n_last + np.arange(len(held))
    last["obs"] = np.concatenate([np.where(rng.random(n_last) < 0.2, 0, rng.integers(1, 9, n_last)), rng.integers(0, 3, len(held))]).astype(np.int32)
    return R, last, cur


@pytest.mark.parametrize("seed,n_left,n_right,n_last,th", [(51, 1500, 1400, 1800, 7.0), (52, 900, 1000, 3000, 15.0), (53, 1200, 0, 1500, 7.0), (54, 0, 900, 1200, 7.0),
                                                           (55, 2000, 1900, 0, 7.0), (56, 600, 700, 4000, 3.0)])
def test_search_by_projection_last_frame_two_camera_tables(msorb_mod, oracle, seed, n_left, n_right, n_last, th):
    """msorb_search_by_projection_frames_rig (ORBmatcher.cc:1941-2152 with the right arm :2059-2124) vs the oracle's arm"""
    R, last, cur = make_last_table(oracle, seed, n_left, n_right, n_last, th)
    ofl, ofr = oracle.OracleFrame(R["kl"], R["dl"], None, BOUNDS, SCALE), oracle.OracleFrame(R["kr"], R["dr"], None, BOUNDS, SCALE)
    dfl, dfr = msorb_mod.Frame(R["kl"], R["dl"], None, BOUNDS, SCALE), msorb_mod.Frame(R["kr"], R["dr"], None, BOUNDS, SCALE)
    try:
        for fwd, bwd, ori in ((False, False, True), (True, False, True), (False, True, False)):
            want, got = cur.copy(), cur.copy()
            wn = oracle.search_by_projection_frames_rig(ofl, ofr, last, want, th, fwd, bwd, ori)
            gn = msorb_mod.search_by_projection_frames_rig(dfl, dfr, last, got, th, fwd, bwd, ori)
            assert gn == wn and np.array_equal(got, want), (seed, fwd, bwd, ori, gn, wn, int((got != want).sum()))
            if n_last >= 1500 and n_left >= 600 and n_right >= 600 and not fwd and not bwd:
                new = (want >= 0) & (want < n_last)
                assert new[:n_left].sum() > 100 and new[n_left:].sum() > 40
    finally:
        dfl.close(); dfr.close()
