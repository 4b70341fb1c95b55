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
         dict(seed=8, n_left=1800, n_right=1700, M=5000, th=2.0)]


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
