"""GPU parity: DBoW2 transform (msorb_bow_transform[_batch]) and ComputeDistinctiveDescriptors
(msorb_distinctive_descriptors) through the C ABI vs oracle/bow_oracle.cc — bit-exact, doubles included."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "ms-slam_amd"), os.path.join(ROOT, "oracle"), os.path.dirname(os.path.abspath(__file__))]
import bow_cases  # noqa: E402

pytestmark = pytest.mark.gpu

KEYS = ("bow_word", "bow_value", "fv_node", "fv_begin", "fv_feat", "feat_word", "feat_node", "feat_weight")


def _pair(voc, scoring=0, weighting=0):
    import msorb
    import orb_oracle
    args = (voc["k"], voc["L"], scoring, weighting, voc["parent"], voc["is_leaf"], voc["descriptors"], voc["weights"])
    return msorb.Vocabulary(*args), orb_oracle.OracleVocabulary(*args)


def _same(a, b, keys=KEYS):
    for k in keys:
        assert a[k].dtype == b[k].dtype and a[k].shape == b[k].shape, k
        assert a[k].tobytes() == b[k].tobytes(), k     # doubles compared by bit pattern


@pytest.mark.parametrize("cfg", [dict(k=10, L=3), dict(k=10, L=4, irregular=True, dfs_ids=True),
                                 dict(k=4, L=5, irregular=True, tie_frac=0.3), dict(k=20, L=2, tie_frac=0.2),
                                 dict(k=2, L=8, irregular=True, stop_frac=0.3)])
def test_transform_matches_oracle(cfg):
    voc = bow_cases.make_vocabulary(11, **cfg)
    dev, orc = _pair(voc)
    try:
        for n in (0, 1, 7, 500, 2017):
            feats = bow_cases.make_features(n, voc, n)
            for levelsup in (4, 0, 1, cfg["L"], cfg["L"] + 2):
                _same(dev.transform(feats, levelsup), orc.transform(feats, levelsup))
    finally:
        dev.close()


@pytest.mark.parametrize("scoring,weighting", [(0, 0), (1, 0), (5, 0), (5, 1), (0, 1), (2, 2), (3, 3), (4, 0)])
def test_weighting_and_scoring_modes(scoring, weighting):
    voc = bow_cases.make_vocabulary(5, k=6, L=3, stop_frac=0.1)
    dev, orc = _pair(voc, scoring, weighting)
    try:
        feats = bow_cases.make_features(9, voc, 1500)     # 216 words, 1500 features: long accumulation runs
        _same(dev.transform(feats), orc.transform(feats))
    finally:
        dev.close()


def test_maximum_frame_size_and_capacity_error():
    import msorb
    voc = bow_cases.make_vocabulary(8, k=10, L=3)
    dev, orc = _pair(voc)
    try:
        feats = bow_cases.make_features(1, voc, 8192)
        _same(dev.transform(feats), orc.transform(feats))
        with pytest.raises(msorb.MsorbError) as e:
            dev.transform(np.zeros((8193, 32), np.uint8))
        assert e.value.code == msorb.E_CAPACITY
    finally:
        dev.close()


def test_empty_vocabulary_and_all_stopped():
    import msorb
    empty = msorb.Vocabulary(10, 6, 0, 0, np.zeros(1, np.int32), np.zeros(1, np.uint8), np.zeros((1, 32), np.uint8),
                             np.zeros(1))
    r = empty.transform(np.zeros((5, 32), np.uint8))
    assert len(r["bow_word"]) == 0 and len(r["fv_node"]) == 0 and r["fv_begin"].tolist() == [0]
    empty.close()
    voc = bow_cases.make_vocabulary(8, k=3, L=2, stop_frac=1.0)
    dev, orc = _pair(voc)
    try:
        feats = bow_cases.make_features(2, voc, 50)
        a = dev.transform(feats)
        _same(a, orc.transform(feats))
        assert len(a["bow_word"]) == 0
    finally:
        dev.close()


def test_batch_on_extractor_outputs_matches_per_frame():
    """transform_batch consumes msorb_extract_batch's device descriptors directly (ragged counts)."""
    import torch
    import msorb
    from msorb import synth
    voc = bow_cases.make_vocabulary(21, k=10, L=4)
    dev, orc = _pair(voc)
    ex = msorb.ORBextractor(1000, 1.2, 8, 20, 7)
    try:
        imgs = synth.stereo_batch(3, 240, 320, seed0=5)          # 6 images
        d_img = torch.from_numpy(imgs).cuda()
        counts, _, d_kps, d_desc = ex.extract_batch(d_img, (0, 0))
        counts = counts.copy()
        counts[1] = 0                                            # an empty frame in the middle
        out = dev.transform_batch(d_desc, counts)
        desc_h = d_desc.cpu().numpy()
        for i, n in enumerate(counts):
            ref = orc.transform(desc_h[i, :n])
            nb, nf = int(out["n_bow"][i]), int(out["n_fv"][i])
            got = dict(bow_word=out["bow_word"][i, :nb].cpu().numpy(), bow_value=out["bow_value"][i, :nb].cpu().numpy(),
                       fv_node=out["fv_node"][i, :nf].cpu().numpy(), fv_begin=out["fv_begin"][i, :nf + 1].cpu().numpy())
            got["fv_feat"] = out["fv_feat"][i, :int(got["fv_begin"][nf])].cpu().numpy()
            _same(got, ref, ("bow_word", "bow_value", "fv_node", "fv_begin", "fv_feat"))
        assert out["elapsed_ms"] > 0
    finally:
        dev.close()
        ex.close()


def test_text_loader_round_trip(tmp_path):
    import msorb
    import orb_oracle
    voc = bow_cases.make_vocabulary(4, k=5, L=3, irregular=True)
    path = tmp_path / "voc.txt"
    feats = bow_cases.make_features(3, voc, 300)
    orc = orb_oracle.OracleVocabulary(voc["k"], voc["L"], 0, 0, voc["parent"], voc["is_leaf"], voc["descriptors"],
                                      voc["weights"])
    for trailing in (True, False):
        bow_cases.write_text(path, voc, trailing_newline=trailing)
        dev = msorb.Vocabulary(path=path)
        try:
            assert (dev.k, dev.L, dev.n_nodes, dev.n_words) == (5, 3, len(voc["parent"]), int(voc["is_leaf"].sum()))
            _same(dev.transform(feats), orc.transform(feats))
        finally:
            dev.close()
    (tmp_path / "bad.txt").write_text("99 6 0 0\n")
    with pytest.raises(msorb.MsorbError):
        msorb.Vocabulary(path=tmp_path / "bad.txt")


def test_distinctive_descriptors_matches_oracle():
    import msorb
    import orb_oracle
    rng = np.random.default_rng(0)
    sizes = [0, 1, 2, 3, 8, 9, 16, 17, 32, 33, 63, 64, 65, 130, 300] + rng.integers(0, 40, 3000).tolist()
    desc, ob = bow_cases.make_observations(7, sizes)
    bi, bm, ms = msorb.distinctive_descriptors(desc, ob)
    ei, em = orb_oracle.distinctive_descriptors(desc, ob)
    assert bi.tolist() == ei.tolist()
    assert bm.tolist() == em.tolist()
    assert ms > 0
    # no points / only empty points
    bi, _, _ = msorb.distinctive_descriptors(np.zeros((0, 32), np.uint8), np.zeros(4, np.int32))
    assert bi.tolist() == [-1, -1, -1]
