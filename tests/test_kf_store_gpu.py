"""Resident KeyFrame store (msorb_kf_store_*): SearchByBoW and SearchForTriangulation with the KeyFrames' descriptors,
keypoints and FeatureVectors kept on the device give exactly what the per-call entries give (which are pinned to the oracle
in test_bow_match.py) — and the oracle is asked directly as well."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "ms-slam_amd"), os.path.join(ROOT, "oracle"), os.path.dirname(os.path.abspath(__file__))]
import bow_match_cases as bmc  # noqa: E402

pytestmark = pytest.mark.gpu
SCALE = np.array([1.2 ** i for i in range(8)], np.float32)
SIGMA2 = (SCALE * SCALE).astype(np.float32)


def _kps(n, angle):
    import orb_oracle
    k = np.zeros(n, orb_oracle.KP_DTYPE)
    k["angle"] = angle
    return k


def test_search_by_bow_resident_equals_per_call_and_oracle():
    import msorb
    import orb_oracle
    K = 12
    cand = [bmc.make_pair(300 + i, n1=1500 + 37 * i, n2=1800, n_nodes=70, mask_frac=0.2) for i in range(K)]
    for p in cand:                                            # one frame, many candidate KeyFrames (relocalisation)
        p["avail2"] = None
        p["desc2"], p["fv2"], p["angle2"] = cand[0]["desc2"], cand[0]["fv2"], cand[0]["angle2"]
    st = msorb.KeyFrameStore()
    try:
        ids = [st.add(_kps(len(p["desc1"]), p["angle1"]), p["desc1"], p["fv1"], SCALE, SIGMA2) for p in cand]
        assert st.count() == K
        frame = dict(desc=cand[0]["desc2"], fv=cand[0]["fv2"], angle=cand[0]["angle2"])
        for th, inc, ratio, ori in ((50, True, 0.7, True), (50, False, 0.8, False)):
            got, ms = st.search_by_bow([dict(kf1=ids[i], kf2=-1, valid1=cand[i]["valid1"]) for i in range(K)], frame, th, inc, ratio, ori)
            want, _ = msorb.search_by_bow(cand, th, inc, ratio, ori)
            assert ms > 0
            for g, w, p in zip(got, want, cand):
                assert g[0] == w[0] and g[1].tolist() == w[1].tolist() and g[2].tolist() == w[2].tolist()
                nm, m12, m21 = orb_oracle.search_by_bow(p["desc1"], p["desc2"], p["valid1"], None, p["fv1"], p["fv2"], p["angle1"],
                                                        p["angle2"], th, inc, ratio, ori)
                assert g[0] == nm and g[1].tolist() == m12.tolist()
            assert got[0][0] > 50
        # KeyFrame-KeyFrame form between stored KeyFrames: set 2 of pair i = the queries of KeyFrame i+1
        pairs, ref = [], []
        for i in range(K - 1):
            a, b = cand[i], cand[i + 1]
            avail = (np.arange(len(b["desc1"])) % 5 != 0).astype(np.uint8)
            pairs.append(dict(kf1=ids[i], kf2=ids[i + 1], valid1=a["valid1"], avail2=avail))
            ref.append(dict(desc1=a["desc1"], desc2=b["desc1"], valid1=a["valid1"], avail2=avail, fv1=a["fv1"], fv2=b["fv1"],
                            angle1=a["angle1"], angle2=b["angle1"]))
        got, _ = st.search_by_bow(pairs, None, 50, False, 0.75, True)
        want, _ = msorb.search_by_bow(ref, 50, False, 0.75, True)
        for g, w in zip(got, want):
            assert g[0] == w[0] and g[1].tolist() == w[1].tolist() and g[2].tolist() == w[2].tolist()
        # a removed KeyFrame is refused, the others keep working; mixing frame / KeyFrame trains in one call is refused
        st.remove(ids[3])
        assert st.count() == K - 1
        with pytest.raises(msorb.MsorbError):
            st.search_by_bow([dict(kf1=ids[3], kf2=-1, valid1=cand[3]["valid1"])], frame)
        with pytest.raises(msorb.MsorbError):
            st.search_by_bow([dict(kf1=ids[0], kf2=ids[1], valid1=cand[0]["valid1"])], frame)
        again, _ = st.search_by_bow([dict(kf1=ids[5], kf2=-1, valid1=cand[5]["valid1"])], frame, 50, True, 0.7, True)
        want5, _ = msorb.search_by_bow([cand[5]], 50, True, 0.7, True)
        assert again[0][1].tolist() == want5[0][1].tolist()
    finally:
        st.close()


def test_search_for_triangulation_resident_equals_per_call():
    import msorb
    import orb_oracle
    tri = [bmc.make_triangulation_pair(400 + i, n1=1600, n2=1500 + 41 * i, n_nodes=60, mask_frac=0.35) for i in range(8)]
    for p in tri:                                             # CreateNewMapPoints: one new KeyFrame against its neighbours
        for k in ("desc1", "fv1", "kp1"):
            p[k] = tri[0][k]
    st = msorb.KeyFrameStore()
    try:
        cur = st.add(tri[0]["kp1"], tri[0]["desc1"], tri[0]["fv1"], SCALE, SIGMA2)
        nb = [st.add(p["kp2"], p["desc2"], p["fv2"], p["scale_factors2"], p["level_sigma2_2"]) for p in tri]
        for coarse, ori in ((False, True), (True, False)):
            got, ms = st.search_for_triangulation([dict(kf1=cur, kf2=nb[i], valid1=p["valid1"], avail2=p["avail2"], stereo1=p["stereo1"],
                                                        stereo2=p["stereo2"], F12=p["F12"], ep=p["ep"]) for i, p in enumerate(tri)],
                                                  coarse, ori)
            want, _ = msorb.search_for_triangulation(tri, coarse, ori)
            for g, w in zip(got, want):
                assert g[0] == w[0] and g[1].tolist() == w[1].tolist()
            assert got[0][0] > 30 and ms > 0
        nm, m12 = orb_oracle.search_for_triangulation(tri[2])
        one, _ = st.search_for_triangulation([dict(kf1=cur, kf2=nb[2], valid1=tri[2]["valid1"], avail2=tri[2]["avail2"],
                                                   stereo1=tri[2]["stereo1"], stereo2=tri[2]["stereo2"], F12=tri[2]["F12"], ep=tri[2]["ep"])])
        assert one[0][0] == nm and one[0][1].tolist() == m12.tolist()
    finally:
        st.close()


def test_removed_keyframes_give_their_rows_and_ids_back():
    """A sequence adds and removes KeyFrames for as long as it runs (culling, sparsification re-adds, evictions): the store's
    footprint must follow the live set.  Rounds of add / remove / add with mixed sizes: rows reserved stay where the first round
    left them, ids are reused, searches on the survivors keep their results, and a search on a removed id is refused."""
    import msorb
    rng = np.random.default_rng(9)
    pairs = [bmc.make_pair(700 + i, n1=600 + 97 * (i % 7), n2=900, n_nodes=40) for i in range(24)]
    frame = dict(desc=pairs[0]["desc2"], fv=pairs[0]["fv2"], angle=pairs[0]["angle2"])
    st = msorb.KeyFrameStore()
    try:
        def add(i):
            p = pairs[i]
            return st.add(_kps(len(p["desc1"]), p["angle1"]), p["desc1"], p["fv1"], SCALE, SIGMA2)

        def search(kid, i):
            (r,), _ = st.search_by_bow([dict(kf1=kid, kf2=-1, valid1=pairs[i]["valid1"])], frame, 50, True, 0.7, True)
            return r[0], r[2].copy()
        live = {i: add(i) for i in range(24)}
        want = {i: search(live[i], i) for i in range(24)}
        used0, reserved0 = st.rows()
        assert used0 == sum(len(pairs[i]["desc1"]) for i in range(24)) and reserved0 >= used0
        max_id = max(live.values())
        for rnd in range(30):
            out = [int(i) for i in rng.choice(sorted(live), 9, replace=False)]
            dead = {i: live.pop(i) for i in out}
            for kid in dead.values():
                st.remove(kid)
            assert st.count() == len(live)
            with pytest.raises(msorb.MsorbError):
                search(dead[out[0]], out[0])                       # a removed id is unknown to the store
            for i in rng.permutation(out):
                live[int(i)] = add(int(i))                         # (another order than they were removed in: other ranges)
            assert max(live.values()) <= max_id, "ids of removed KeyFrames were not reused"
            used, reserved = st.rows()
            assert used == used0 and reserved == reserved0, f"round {rnd}: {used} rows in use, {reserved} reserved (was {reserved0})"
            for i in (out[0], out[-1], sorted(live)[0]):
                n, m21 = search(live[i], i)
                assert n == want[i][0] and np.array_equal(m21, want[i][1])
        for kid in live.values():
            st.remove(kid)
        assert st.count() == 0 and st.rows()[0] == 0
    finally:
        st.close()
