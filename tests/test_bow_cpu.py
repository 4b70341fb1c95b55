"""CPU checks of oracle/bow_oracle.cc (DBoW2 transform + ComputeDistinctiveDescriptors restatement) against
hand-computed answers and an independent numpy formulation."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "oracle"), os.path.dirname(os.path.abspath(__file__))]
import bow_cases  # noqa: E402
import orb_oracle  # noqa: E402


def _d(*bytes_):
    a = np.zeros(32, np.uint8)
    a[:len(bytes_)] = bytes_
    return a


def _tiny():
    """k=2, L=2: root -> A(1), B(2); A -> words 0 (node 3), 1 (node 4); B -> words 2 (node 5), 3 (node 6)."""
    desc = np.stack([_d(), _d(0x00), _d(0xFF, 0xFF), _d(0x01), _d(0x0E), _d(0xFF, 0xF0), _d(0xFF, 0xFF, 0xFF)])
    parent = np.array([0, 0, 0, 1, 1, 2, 2], np.int32)
    leaf = np.array([0, 0, 0, 1, 1, 1, 1], np.uint8)
    w = np.array([0, 0, 0, 1.5, 0.25, 2.0, 0.0])
    return parent, leaf, desc, w


def test_transform_hand_computed():
    parent, leaf, desc, w = _tiny()
    voc = orb_oracle.OracleVocabulary(2, 2, 0, 0, parent, leaf, desc, w)   # L1 norm, TF-IDF
    feats = np.stack([_d(0x01),              # A, word 0
                      _d(0x0F),              # A (4 vs 12 bits), word 1 (1 vs 3 bits)
                      _d(0x03),              # A, tie between words (1 bit each) -> first child = word 0
                      _d(0xFF, 0xFF, 0xFF),  # B, word 3: weight 0 -> stopped, in neither container
                      _d(0xFF, 0xF1),        # B, word 2
                      _d(0x01)])             # word 0 again
    r = voc.transform(feats, levelsup=1)     # node ids at level L-1 = 1
    assert r["feat_word"].tolist() == [0, 1, 0, 3, 2, 0]
    assert r["feat_node"].tolist() == [1, 1, 1, 2, 2, 1]
    assert r["bow_word"].tolist() == [0, 1, 2]
    acc = np.array([(1.5 + 1.5) + 1.5, 0.25, 2.0])
    norm = (0.0 + acc[0]) + acc[1] + acc[2]
    assert r["bow_value"].tolist() == (acc / norm).tolist()
    assert r["fv_node"].tolist() == [1, 2]
    assert r["fv_begin"].tolist() == [0, 4, 5]
    assert r["fv_feat"].tolist() == [0, 1, 2, 5, 4]
    # levelsup >= L: every feature files under the root (nid_level <= 0, TemplatedVocabulary.h:1228)
    r0 = voc.transform(feats, levelsup=2)
    assert r0["fv_node"].tolist() == [0] and r0["fv_feat"].tolist() == [0, 1, 2, 4, 5]
    # levelsup = 0: leaves themselves
    r2 = voc.transform(feats, levelsup=0)
    assert r2["fv_node"].tolist() == [3, 4, 5] and r2["fv_begin"].tolist() == [0, 3, 4, 5]


def test_transform_weighting_and_scoring_modes():
    parent, leaf, desc, w = _tiny()
    feats = np.stack([_d(0x01), _d(0x01), _d(0x0F), _d(0xFF, 0xF1)])
    # BINARY/IDF weighting: addIfNotExist -> one weight per word; DOT_PRODUCT (5) does not normalise
    r = orb_oracle.OracleVocabulary(2, 2, 5, 2, parent, leaf, desc, w).transform(feats, 1)
    assert r["bow_value"].tolist() == [1.5, 0.25, 2.0]
    # TF weighting without normalisation: divided by the number of words (TemplatedVocabulary.h:1162-1168)
    r = orb_oracle.OracleVocabulary(2, 2, 5, 1, parent, leaf, desc, w).transform(feats, 1)
    assert r["bow_value"].tolist() == [(1.5 + 1.5) / 3.0, 0.25 / 3.0, 2.0 / 3.0]
    # L2 scoring
    r = orb_oracle.OracleVocabulary(2, 2, 1, 0, parent, leaf, desc, w).transform(feats, 1)
    nrm = np.sqrt(3.0 * 3.0 + 0.25 * 0.25 + 2.0 * 2.0)
    np.testing.assert_allclose(r["bow_value"], np.array([3.0, 0.25, 2.0]) / nrm, rtol=1e-15)


def test_shallow_leaf_keeps_previous_node_id():
    # root -> leaf X (node 1, word 0) and inner Y (node 2) -> words 1, 2.  With levelsup=0 (nid_level=2) a feature that
    # stops at X never reaches level 2: documented convention = previous feature's node id (0 for the first).
    desc = np.stack([_d(), _d(0x00), _d(0xFF), _d(0xF0), _d(0xFF)])
    parent = np.array([0, 0, 0, 2, 2], np.int32)
    leaf = np.array([0, 1, 0, 1, 1], np.uint8)
    w = np.array([0, 1.0, 0, 2.0, 3.0])
    voc = orb_oracle.OracleVocabulary(2, 2, 0, 0, parent, leaf, desc, w)
    r = voc.transform(np.stack([_d(0x00), _d(0xFF), _d(0x01)]), levelsup=0)
    assert r["feat_word"].tolist() == [0, 2, 0]
    assert r["feat_node"].tolist() == [0, 4, 4]


def _numpy_distinctive(desc, ob):
    best, med = [], []
    bits = np.unpackbits(desc, axis=1).astype(np.int32)
    for p in range(len(ob) - 1):
        b = bits[ob[p]:ob[p + 1]]
        n = len(b)
        if n == 0:
            best.append(-1)
            med.append(np.iinfo(np.int32).max)
            continue
        dm = (b[:, None, :] != b[None, :, :]).sum(-1)
        m = np.sort(dm, axis=1)[:, int(0.5 * (n - 1))]
        best.append(int(np.argmin(m)))      # first minimum
        med.append(int(m.min()))
    return np.array(best), np.array(med)


def test_distinctive_descriptors_vs_numpy():
    sizes = [0, 1, 2, 3, 4, 5, 8, 9, 16, 17, 31, 33, 64, 65, 100]
    desc, ob = bow_cases.make_observations(3, sizes)
    bi, bm = orb_oracle.distinctive_descriptors(desc, ob)
    ei, em = _numpy_distinctive(desc, ob)
    assert bi.tolist() == ei.tolist()
    assert bm.tolist() == em.tolist()


def test_case_generator_shapes():
    voc = bow_cases.make_vocabulary(1, k=10, L=3)
    assert len(voc["parent"]) == 1 + 10 + 100 + 1000 and int(voc["is_leaf"].sum()) == 1000
    irr = bow_cases.make_vocabulary(2, k=6, L=4, irregular=True, dfs_ids=True)
    assert (irr["parent"][1:] < np.arange(1, len(irr["parent"]))).all()
