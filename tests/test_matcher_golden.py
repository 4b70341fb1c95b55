"""tests/golden/matchers_golden.json (tools/make_golden_matchers.py): the oracle (CPU) and the device entry points (GPU)
against the committed digests of the matcher-family rows."""
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "ms-slam_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tools"),
                os.path.dirname(os.path.abspath(__file__))]
GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "matchers_golden.json")))


def test_oracle_matches_committed_matcher_goldens():
    import make_golden_matchers
    now = make_golden_matchers.compute()
    for row, want in GOLD.items():
        assert now[row]["inputs"] == want["inputs"], f"{row}: the seeded input generator drifted"
        assert now[row] == want, row


@pytest.mark.gpu
def test_device_matches_committed_matcher_goldens():
    import msorb
    import bow_cases
    import bow_match_cases as bmc
    from make_golden_matchers import digest
    p = bmc.make_pair(5, n1=700, n2=800, n_nodes=30)
    (a,), _ = msorb.search_by_bow([p], 50, True, 0.7, True)
    (b,), _ = msorb.search_by_bow([p], 50, False, 0.8, False)
    assert [a[0], b[0]] == GOLD["search_by_bow"]["nmatches"]
    assert digest(a[1], a[2], b[1], b[2]) == GOLD["search_by_bow"]["outputs"]
    t = bmc.make_triangulation_pair(6, n1=700, n2=800, n_nodes=25)
    (a,), _ = msorb.search_for_triangulation([t], False, True)
    (b,), _ = msorb.search_for_triangulation([t], True, False)
    assert [a[0], b[0]] == GOLD["search_for_triangulation"]["nmatches"]
    assert digest(a[1], b[1]) == GOLD["search_for_triangulation"]["outputs"]
    voc = bow_cases.make_vocabulary(4, k=8, L=4, irregular=True, stop_frac=0.05)
    feats = bow_cases.make_features(3, voc, 1200)
    dev = msorb.Vocabulary(voc["k"], voc["L"], 0, 0, voc["parent"], voc["is_leaf"], voc["descriptors"], voc["weights"])
    try:
        r = dev.transform(feats, 2)
    finally:
        dev.close()
    assert len(r["bow_word"]) == GOLD["bow_transform"]["n_words"]
    assert digest(r["bow_word"], r["bow_value"], r["fv_node"], r["fv_begin"], r["fv_feat"]) == GOLD["bow_transform"]["outputs"]
    obs, ob = bow_cases.make_observations(9, [0, 1, 2, 7, 8, 9, 30, 64, 65, 100] + list(range(2, 40)))
    bi, bm, _ = msorb.distinctive_descriptors(obs, ob)
    assert digest(bi, bm) == GOLD["distinctive_descriptors"]["outputs"]
