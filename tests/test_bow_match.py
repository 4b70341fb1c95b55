"""ORBmatcher::SearchByBoW (ORBmatcher.cc:223-421, 872-1166): the C++ oracle against a definition-level restatement
(CPU), and msorb_search_by_bow through the C ABI against the oracle (GPU), bit-exact."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "ms-slam_amd"), os.path.join(ROOT, "oracle"), os.path.dirname(os.path.abspath(__file__))]
import bow_match_cases as bmc  # noqa: E402

MODES = [(50, True, 0.7, True), (50, False, 0.8, True), (50, True, 0.75, False), (100, True, 0.9, True)]


def _oracle(p, th, inc, ratio, ori):
    import orb_oracle
    return orb_oracle.search_by_bow(p["desc1"], p["desc2"], p["valid1"], p["avail2"], p["fv1"], p["fv2"], p["angle1"],
                                    p["angle2"], th, inc, ratio, ori)


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_oracle_matches_definition(seed):
    p = bmc.make_pair(seed, n1=260, n2=300, n_nodes=12, shuffle_lists=seed == 2)
    if seed == 1:
        p["avail2"] = None
    for th, inc, ratio, ori in MODES:
        nm, m12, m21 = _oracle(p, th, inc, ratio, ori)
        wn, w12 = bmc.naive(p, th, inc, ratio, ori)
        assert nm == wn and m12.tolist() == w12.tolist()
        inv = -np.ones(len(p["desc2"]), np.int64)
        inv[m12[m12 >= 0]] = np.nonzero(m12 >= 0)[0]
        assert m21.tolist() == inv.tolist()
        assert nm > 20


CASES = [dict(n1=900, n2=1100, n_nodes=60), dict(n1=2000, n2=2000, n_nodes=100, shuffle_lists=True),
         dict(n1=700, n2=1500, n_nodes=3), dict(n1=300, n2=5000, n_nodes=1, mask_frac=0.5),
         dict(n1=50, n2=0, n_nodes=4), dict(n1=0, n2=40, n_nodes=4), dict(n1=400, n2=400, n_nodes=400, dup_frac=0.5)]


@pytest.mark.gpu
@pytest.mark.parametrize("case", range(len(CASES)))
def test_device_matches_oracle(case):
    import msorb
    p = bmc.make_pair(100 + case, **CASES[case])
    for k, (th, inc, ratio, ori) in enumerate(MODES):
        q = dict(p)
        if k == 2:
            q["avail2"] = None
        (got,), ms = msorb.search_by_bow([q], th, inc, ratio, ori)
        nm, m12, m21 = _oracle(q, th, inc, ratio, ori)
        assert got[0] == nm
        assert got[1].tolist() == m12.tolist()
        assert got[2].tolist() == m21.tolist()
    if CASES[case]["n1"] >= 300 and CASES[case]["n2"] >= 400:
        assert nm > 10


@pytest.mark.gpu
def test_batch_equals_singles_and_rejects_bad_vectors():
    import msorb
    pairs = [bmc.make_pair(7 + i, n1=300 + 211 * i, n2=1700 - 190 * i, n_nodes=5 + 17 * i) for i in range(6)]
    pairs.append(bmc.make_pair(99, n1=0, n2=0))
    res, ms = msorb.search_by_bow(pairs, 50, True, 0.7, True)
    assert ms > 0
    for p, r in zip(pairs, res):
        nm, m12, m21 = _oracle(p, 50, True, 0.7, True)
        assert r[0] == nm and r[1].tolist() == m12.tolist() and r[2].tolist() == m21.tolist()
    bad = dict(pairs[0])
    f = [a.copy() for a in bad["fv2"]]
    f[2][1] = f[2][0]                                  # a feature listed twice
    bad["fv2"] = tuple(f)
    with pytest.raises(msorb.MsorbError):
        msorb.search_by_bow([bad])
    bad = dict(pairs[0])
    f = [a.copy() for a in bad["fv1"]]
    f[0][1] = f[0][0]                                  # node ids not strictly ascending
    bad["fv1"] = tuple(f)
    with pytest.raises(msorb.MsorbError):
        msorb.search_by_bow([bad])


# ---- SearchForTriangulation (ORBmatcher.cc:1168-1402) ----
@pytest.mark.parametrize("seed", [0, 1])
def test_triangulation_oracle_matches_definition(seed):
    import orb_oracle
    p = bmc.make_triangulation_pair(seed, n1=300, n2=330, n_nodes=10)
    for coarse, ori in ((False, True), (True, True), (False, False)):
        nm, m12 = orb_oracle.search_for_triangulation(p, coarse, ori)
        wn, w12 = bmc.naive_triangulation(p, coarse, ori)
        assert nm == wn and m12.tolist() == w12.tolist()
        assert nm > 15
    # the gates must bite: the fine search keeps fewer pairs than the coarse one
    assert orb_oracle.search_for_triangulation(p, False, False)[0] < orb_oracle.search_for_triangulation(p, True, False)[0]


TCASES = [dict(n1=900, n2=1000, n_nodes=40), dict(n1=2000, n2=2000, n_nodes=100, pix_noise=0.7),
          dict(n1=600, n2=1500, n_nodes=2), dict(n1=200, n2=4000, n_nodes=1, mask_frac=0.5), dict(n1=30, n2=0, n_nodes=3),
          dict(n1=0, n2=30, n_nodes=3)]


@pytest.mark.gpu
@pytest.mark.parametrize("case", range(len(TCASES)))
def test_triangulation_device_matches_oracle(case):
    import msorb
    import orb_oracle
    p = bmc.make_triangulation_pair(200 + case, **TCASES[case])
    for coarse, ori in ((False, True), (True, True), (False, False)):
        (got,), ms = msorb.search_for_triangulation([p], coarse, ori)
        nm, m12 = orb_oracle.search_for_triangulation(p, coarse, ori)
        assert got[0] == nm and got[1].tolist() == m12.tolist()
    if TCASES[case]["n1"] >= 200 and TCASES[case]["n2"] >= 1000:
        assert nm > 10


@pytest.mark.gpu
def test_triangulation_batch_equals_singles():
    import msorb
    import orb_oracle
    pairs = [bmc.make_triangulation_pair(300 + i, n1=500 + 300 * i, n2=1900 - 250 * i, n_nodes=8 + 20 * i) for i in range(5)]
    res, ms = msorb.search_for_triangulation(pairs)
    assert ms > 0
    for p, r in zip(pairs, res):
        nm, m12 = orb_oracle.search_for_triangulation(p)
        assert r[0] == nm and r[1].tolist() == m12.tolist()
    bad = dict(pairs[0])
    bad["kp2"] = bad["kp2"].copy()
    bad["kp2"]["octave"][3] = 8                          # octave outside pKF2->mvScaleFactors
    with pytest.raises(msorb.MsorbError):
        msorb.search_for_triangulation([bad])
