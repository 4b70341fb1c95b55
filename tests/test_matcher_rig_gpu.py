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
