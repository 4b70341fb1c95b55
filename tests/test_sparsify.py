"""MapSparsification constraint-matrix assembly: oracle properties (CPU) and device parity (GPU)."""
import numpy as np
import pytest

import sparsify_cases as sc

KEYS = ("col_point", "obj_coef", "row_begin", "row_kind", "row_owner", "row_rhs", "col_idx")


def test_oracle_matrix_properties(oracle):
    w = sc.window(1, n_window=10, n_outside=30, n_points=2000, slots_per_kf=600)
    m = oracle.visibility_csr(N=100, **w)
    sp = w["slot_point"]
    valid = sp[sp >= 0]
    # columns: unique valid points in first-encounter order
    _, first = np.unique(valid, return_index=True)
    assert np.array_equal(m["col_point"], valid[np.sort(first)])
    assert m["n_max_obs"] == w["point_nobs"][valid].max()
    assert np.array_equal(m["obj_coef"], (m["n_max_obs"] - w["point_nobs"][m["col_point"]]).astype(np.float32))
    kinds = m["row_kind"]
    assert (kinds == 1).sum() == 10                                   # one row per window keyframe, rhs N
    assert np.all(m["row_rhs"][kinds == 1] == 100) and np.all(m["row_rhs"][kinds == 0] == 1)
    # keyframe row = concatenation of its cell rows
    r = 0
    for k in range(10):
        cells = []
        while kinds[r] == 0:
            cells.append(m["col_idx"][m["row_begin"][r]:m["row_begin"][r + 1]]); r += 1
        kf = m["col_idx"][m["row_begin"][r]:m["row_begin"][r + 1]]
        assert np.array_equal(np.concatenate(cells) if cells else np.zeros(0, np.int32), kf)
        assert len(kf) == (sp[w["kf_slot_begin"][k]:w["kf_slot_begin"][k + 1]] >= 0).sum()
        r += 1
    # outside rows: ascending keyframe id, rhs = count/total*N in float
    ext = np.nonzero(kinds == 2)[0]
    assert np.all(np.diff(m["row_owner"][ext]) > 0) and not w["kf_in_window"][m["row_owner"][ext]].any()
    for r in ext[:20]:
        cols = m["col_idx"][m["row_begin"][r]:m["row_begin"][r + 1]]
        kf = m["row_owner"][r]
        assert np.all(np.diff(cols) > 0)
        want = [c for c, p in enumerate(m["col_point"]) if kf in w["obs_kf"][w["obs_begin"][p]:w["obs_begin"][p + 1]]]
        assert cols.tolist() == want
        assert m["row_rhs"][r] == np.float32(np.float32(len(cols)) / np.float32(w["kf_num_mps"][kf])) * np.float32(100)


@pytest.mark.gpu
@pytest.mark.parametrize("seed,kw", [(1, {}), (2, dict(n_window=3, n_outside=0, n_points=500, slots_per_kf=300)),
                                     (3, dict(n_window=30, n_outside=200, n_points=20000, slots_per_kf=2000, tracked_frac=0.9)),
                                     (4, dict(n_window=1, n_outside=5, n_points=50, slots_per_kf=40, tracked_frac=0.0))])
def test_device_matrix_matches_oracle(msorb_mod, oracle, seed, kw):
    w = sc.window(seed, **kw)
    got = msorb_mod.visibility_csr(N=100, **w)
    want = oracle.visibility_csr(N=100, **w)
    assert (got["n_cols"], got["n_rows"], got["n_max_obs"]) == (want["n_cols"], want["n_rows"], want["n_max_obs"])
    for k in KEYS:
        assert np.array_equal(got[k], want[k]), k
