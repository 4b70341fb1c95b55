"""CPU-only tests pinning the matcher oracle to the definitions it restates."""
import numpy as np

from msorb import synth
import matcher_cases as mc


def test_descriptor_distance_is_popcount_of_xor(oracle):
    rng = np.random.Generator(np.random.PCG64(1))
    a = rng.integers(0, 256, (200, 32), dtype=np.uint8)
    b = rng.integers(0, 256, (200, 32), dtype=np.uint8)
    a[0] = 0; b[0] = 255; a[1] = b[1]; b[2] = a[2]; b[2, 31] ^= 0x80
    for i in range(200):
        assert oracle.descriptor_distance(a[i], b[i]) == int(np.unpackbits(a[i] ^ b[i]).sum())
    assert oracle.descriptor_distance(a[0], b[0]) == 256 and oracle.descriptor_distance(a[1], b[1]) == 0
    assert oracle.descriptor_distance(a[2], b[2]) == 1


def test_three_maxima_known_cases(oracle, msorb_mod):
    cases = [([0] * 30, [-1, -1, -1]), ([5] + [0] * 29, [0, -1, -1]), ([10, 9, 8] + [0] * 27, [0, 1, 2]),
             ([100, 9, 8] + [0] * 27, [0, -1, -1]), ([100, 50, 9] + [0] * 27, [0, 1, -1]), ([3, 3, 3, 3] + [0] * 26, [0, 1, 2])]
    for sizes, want in cases:
        assert oracle.three_maxima(sizes).tolist() == want
        assert msorb_mod.three_maxima(sizes).tolist() == want      # host-only product code path


def test_features_in_area_matches_brute_force_set(oracle):
    img = synth.image(3, 376, 1241)
    ex = oracle.OracleExtractor(2000, 1.2, 8, 20, 7)
    _, kps, desc = ex(img)
    f = oracle.OracleFrame(kps, desc, None, (0.0, 1241.0, 0.0, 376.0), ex.tables()["scale"])
    rng = np.random.Generator(np.random.PCG64(2))
    for _ in range(200):
        x, y, r = rng.uniform(0, 1241), rng.uniform(0, 376), rng.uniform(2, 60)
        got = f.GetFeaturesInArea(x, y, r)
        brute = np.nonzero((np.abs(kps["x"] - np.float32(x)) < np.float32(r)) & (np.abs(kps["y"] - np.float32(y)) < np.float32(r)))[0]
        # the grid walk may only lose points whose rounded cell lies outside the floor/ceil window; with
        # r >= 2 px and 19x8 px cells that never happens for interior points
        assert set(got.tolist()) <= set(brute.tolist())
        assert len(set(got.tolist())) == len(got)
        lv = f.GetFeaturesInArea(x, y, r, 2, 3)
        assert all(2 <= kps["octave"][i] <= 3 for i in lv)


def test_sequential_best_second_equals_sorted_pair(oracle):
    """the strict-'<' scan's (best, second) equals the first two of a (dist, position) sort."""
    rng = np.random.Generator(np.random.PCG64(4))
    for _ in range(500):
        d = rng.integers(0, 6, rng.integers(1, 12))
        best = second = 256
        bi = si = -1
        for i, v in enumerate(d):
            if v < best:
                second, si = best, bi
                best, bi = v, i
            elif v < second:
                second, si = v, i
        order = sorted(range(len(d)), key=lambda i: (d[i], i))
        assert bi == order[0] and best == d[order[0]]
        if len(d) > 1:
            assert si == order[1] and second == d[order[1]]


def test_search_by_projection_claims_are_sequential(oracle):
    """A keypoint claimed by a map point with observations is invisible to later map points
    (ORBmatcher.cc:88-90,129): a second pass can never steal a claimed keypoint."""
    img = synth.image(8, 376, 1241)
    ex = oracle.OracleExtractor(2000, 1.2, 8, 20, 7)
    _, kps, desc = ex(img)
    scale = ex.tables()["scale"]
    f = oracle.OracleFrame(kps, desc, None, (0.0, 1241.0, 0.0, 376.0), scale)
    rng = np.random.Generator(np.random.PCG64(6))
    mp = mc.map_point_table(rng, kps, desc, np.full(len(kps), -1, np.float32), scale, 3000, obs_zero_frac=0.0, sparsified_frac=0.0)
    fm = np.full(len(kps), -1, np.int32)
    n1 = f.SearchByProjection_mps(mp, fm, 3.0)
    assert n1 > 300 and (fm >= 0).sum() == n1          # every accepted map point owns a distinct keypoint
    fm2 = fm.copy()
    n2 = f.SearchByProjection_mps(mp, fm2, 3.0)
    # second pass: claimed keypoints are invisible (all obs > 0), so no owner changes; only free keypoints fill
    assert np.array_equal(fm2[fm >= 0], fm[fm >= 0]) and (fm2 >= 0).sum() == n1 + n2
