"""Compiles the drop-in C++ class ORB_SLAM3::ORBextractor (ms-slam_amd/host, same declaration as the
reference's include/ORBextractor.h) against the test-only cv stub, links libmsorb.so, runs it like
Frame::ExtractORB (Frame.cc:418-425) and compares with the oracle."""
import os
import struct
import subprocess

import numpy as np
import pytest

from msorb import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def test_cpp_dropin_matches_oracle(tmp_path, oracle, msorb_mod):
    exe = tmp_path / "dropin_extractor"
    subprocess.check_call(["g++", "-std=c++17", "-O2", f"-I{ROOT}/tests/cv_stub", f"-I{ROOT}/ms-slam_amd/host",
                           f"-I{ROOT}/include", f"{ROOT}/tests/dropin_extractor_main.cc",
                           f"{ROOT}/ms-slam_amd/host/ORBextractor.cc", f"-L{ROOT}/ms-slam_amd", "-lmsorb",
                           f"-Wl,-rpath,{ROOT}/ms-slam_amd", "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib", "-o", str(exe)])
    cfg = synth.EUROC
    img = synth.image(77, cfg["rows"], cfg["cols"])
    raw, out = tmp_path / "in.raw", tmp_path / "out.bin"
    img.tofile(raw)
    subprocess.check_call([str(exe), str(cfg["rows"]), str(cfg["cols"]), str(raw), str(out), "1000"])
    blob = out.read_bytes()
    mono, n = struct.unpack_from("<ii", blob, 0)
    kps = np.frombuffer(blob, oracle.KP_DTYPE, n, 8)
    desc = np.frombuffer(blob, np.uint8, n * 32, 8 + 28 * n).reshape(n, 32)
    l7r, l7c, lv = struct.unpack_from("<iii", blob, 8 + 60 * n)
    rmono, rkps, rdesc = oracle.OracleExtractor(1000, 1.2, 8, 20, 7)(img)
    assert (mono, n) == (rmono, len(rkps)) and (l7r, l7c, lv) == (134, 210, 8)
    assert np.array_equal(kps.view(np.uint8), rkps.view(np.uint8)) and np.array_equal(desc, rdesc)
    # mvImagePyramid (filled by the asynchronous copy that overlaps the extraction) holds the oracle's levels
    orc = oracle.OracleExtractor(1000, 1.2, 8, 20, 7)
    orc(img)
    pos = 8 + 60 * n + 12
    for l in range(8):
        r, c = struct.unpack_from("<ii", blob, pos)
        want = orc.level(l)
        assert (r, c) == want.shape
        assert np.array_equal(np.frombuffer(blob, np.uint8, r * c, pos + 8).reshape(r, c), want), l
        pos += 8 + r * c
    assert pos == len(blob)


def test_cpp_matcher_adapter_matches_oracle(tmp_path, oracle, msorb_mod):
    """ms-slam_amd/host/ORBmatcher_device.h (the body a maintainer puts into ORBmatcher::SearchByProjection) compiled
    against stand-in Frame / MapPoint types, vs the oracle's restatement of ORBmatcher.cc:43-142."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import matcher_cases as mc
    exe = tmp_path / "dropin_matcher"
    subprocess.check_call(["g++", "-std=c++17", "-O2", f"-I{ROOT}/tests/cv_stub", f"-I{ROOT}/ms-slam_amd/host",
                           f"-I{ROOT}/include", f"{ROOT}/tests/dropin_matcher_main.cc", f"-L{ROOT}/ms-slam_amd", "-lmsorb",
                           f"-Wl,-rpath,{ROOT}/ms-slam_amd", "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib", "-o", str(exe)])
    cfg = synth.KITTI
    ex = msorb_mod.ORBextractor(cfg["nfeatures"], cfg["scale"], cfg["nlevels"], cfg["ini_th"], cfg["min_th"])
    try:
        _, kps, desc = ex(synth.image(5, cfg["rows"], cfg["cols"]))
        scale = ex.GetScaleFactors()
    finally:
        ex.close()
    rng = np.random.Generator(np.random.PCG64(9))
    N, M = len(kps), 3000
    ur = np.where(rng.random(N) < 0.6, kps["x"] - rng.uniform(1, 40, N), -1).astype(np.float32)
    mp = mc.map_point_table(rng, kps, desc, ur, scale, M, 0.2, 0.1)
    init = np.where(rng.random(N) < 0.25, rng.integers(0, M, N), -1).astype(np.int32)
    bounds = (0.0, float(cfg["cols"]), 0.0, float(cfg["rows"]))
    th, th_far, nnratio, far = 3.0, 60.0, 0.8, 1
    blob = tmp_path / "in.bin"
    with open(blob, "wb") as f:
        f.write(struct.pack("<iiii", N, len(scale), M, far))
        f.write(struct.pack("<7f", *bounds, th, th_far, nnratio))
        for a in (kps, desc, ur, np.asarray(scale, np.float32), mp["track_in_view"], mp["bad"], mp["sparsified"],
                  mp["proj_x"], mp["proj_y"], mp["proj_xr"], mp["track_depth"], mp["view_cos"], mp["level"], mp["obs"],
                  mp["desc"], init):
            f.write(np.ascontiguousarray(a).tobytes())
    out = tmp_path / "out.bin"
    subprocess.check_call([str(exe), str(blob), str(out)])
    res = np.frombuffer(out.read_bytes(), np.int32)
    nmatches, got = int(res[0]), res[1:]
    # oracle: the points the C++ side left out of the local map (held by the frame, odd index) are not visited
    held = np.zeros(M, bool)
    held[init[init >= 0]] = True
    mp_o = dict(mp)
    mp_o["track_in_view"] = np.where(held & (np.arange(M) % 2 == 1), 0, mp["track_in_view"]).astype(np.uint8)
    rf = oracle.OracleFrame(kps, desc, ur, bounds, scale)
    want = init.copy()
    rn = rf.SearchByProjection_mps(mp_o, want, th, bFarPoints=bool(far), thFarPoints=th_far, nnratio=nnratio)
    assert rn > 200
    assert nmatches == rn and np.array_equal(got, want)


def test_cpp_sparsification_adapter_matches_oracle(tmp_path, oracle, msorb_mod):
    """ms-slam_amd/host/MapSparsification_device.h over stand-in KeyFrame / MapPoint objects vs the oracle on the flat
    arrays the object graph was built from (outside-keyframe rows compared by owner: their order is run dependent in
    the reference, a std::map keyed by shared_ptr)."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import sparsify_cases as sc
    exe = tmp_path / "dropin_sparsify"
    subprocess.check_call(["g++", "-std=c++17", "-O2", f"-I{ROOT}/ms-slam_amd/host", f"-I{ROOT}/include",
                           f"{ROOT}/tests/dropin_sparsify_main.cc", f"-L{ROOT}/ms-slam_amd", "-lmsorb",
                           f"-Wl,-rpath,{ROOT}/ms-slam_amd", "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib", "-o", str(exe)])
    w = sc.window(5, n_window=12, n_outside=40, n_points=3000, slots_per_kf=800)
    N = 100
    window_ids = np.nonzero(w["kf_in_window"])[0].astype(np.int32)
    # the generator folds isBad() into slot_point = -1; the object graph needs the held point + a bad flag: mark 3 % of the
    # tracked slots' points bad afterwards and rebuild the flat view the oracle sees
    rng = np.random.Generator(np.random.PCG64(3))
    slot_true = w["slot_point"].copy()
    bad = np.zeros(len(w["point_nobs"]), np.uint8)
    held = np.unique(slot_true[slot_true >= 0])
    bad[rng.choice(held, len(held) // 30, replace=False)] = 1
    flat = dict(w)
    flat["slot_point"] = np.where((slot_true >= 0) & (bad[np.maximum(slot_true, 0)] == 0), slot_true, -1).astype(np.int32)
    valid_pts = flat["slot_point"][flat["slot_point"] >= 0]
    blob, out = tmp_path / "in.bin", tmp_path / "out.bin"
    with open(blob, "wb") as f:
        f.write(struct.pack("<7i", len(window_ids), len(slot_true), len(bad), len(w["obs_kf"]), len(w["kf_in_window"]), N, 48))
        for a in (w["kf_slot_begin"], flat["slot_point"], w["slot_cell"], w["point_nobs"], w["obs_begin"], w["obs_kf"],
                  w["kf_num_mps"], window_ids, slot_true, bad):
            f.write(np.ascontiguousarray(a).tobytes())
    subprocess.check_call([str(exe), str(blob), str(out)])
    raw = out.read_bytes()
    n_cols, n_rows, nnz, nmax = struct.unpack_from("<4i", raw, 0)
    off = 16

    def take(dt, n):
        nonlocal off
        a = np.frombuffer(raw, dt, n, off)
        off += a.nbytes
        return a
    col_point, obj = take(np.int32, n_cols), take(np.float32, n_cols)
    row_begin, row_kind, row_kf, row_cell = take(np.int32, n_rows + 1), take(np.int32, n_rows), take(np.int32, n_rows), take(np.int32, n_rows)
    row_rhs, col_idx = take(np.float32, n_rows), take(np.int32, nnz)
    assert take(np.int32, 1)[0] == 1                                      # side effects
    # bad-but-held points still count for nMaxObsevation? no: :70 skips isBad() — the floor only sees valid ones
    floor = int(w["point_nobs"][valid_pts].max())
    want = oracle.visibility_csr(N=N, n_max_obs_floor=floor, **flat)
    assert (n_cols, nmax) == (want["n_cols"], want["n_max_obs"]) and n_rows == want["n_rows"]
    assert np.array_equal(col_point, want["col_point"]) and np.array_equal(obj, want["obj_coef"])
    inner = want["row_kind"] != 2
    n_in = int(inner.sum())
    assert np.array_equal(row_kind[:n_in], want["row_kind"][:n_in]) and np.array_equal(row_rhs[:n_in], want["row_rhs"][:n_in])
    assert np.array_equal(row_begin[:n_in + 1], want["row_begin"][:n_in + 1])
    assert np.array_equal(col_idx[:row_begin[n_in]], want["col_idx"][:row_begin[n_in]])
    k1 = row_kind[:n_in] == 1
    assert np.array_equal(row_kf[:n_in][k1], window_ids[want["row_owner"][:n_in][k1]])
    assert np.array_equal(row_cell[:n_in][~k1], want["row_owner"][:n_in][~k1])

    def outside(kind, owner, rb, rhs, ci):
        return {int(owner[r]): (float(rhs[r]), ci[rb[r]:rb[r + 1]].tolist()) for r in range(len(kind)) if kind[r] == 2}
    assert outside(row_kind, row_kf, row_begin, row_rhs, col_idx) == \
        outside(want["row_kind"], want["row_owner"], want["row_begin"], want["row_rhs"], want["col_idx"])


def test_cpp_search_local_points_prepass_matches_oracle(tmp_path, oracle, msorb_mod):
    """msorb_host::SearchLocalPointsPrepass (the isInFrustum loop of Tracking::SearchLocalPoints over stand-in Frame /
    MapPoint objects) vs the oracle's Frame::isInFrustum restatement; skipped points must stay untouched."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import frustum_cases as fc
    exe = tmp_path / "dropin_matcher"
    subprocess.check_call(["g++", "-std=c++17", "-O2", f"-I{ROOT}/tests/cv_stub", f"-I{ROOT}/ms-slam_amd/host",
                           f"-I{ROOT}/include", f"{ROOT}/tests/dropin_matcher_main.cc", f"-L{ROOT}/ms-slam_amd", "-lmsorb",
                           f"-Wl,-rpath,{ROOT}/ms-slam_amd", "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib", "-o", str(exe)])
    M, nlev = 5000, 8
    R, t, Ow = fc.pose(4)
    P, Nn, maxd, mind = fc.points(4, M, R, t, Ow)
    rng = np.random.Generator(np.random.PCG64(2))
    seen = (rng.random(M) < 0.2).astype(np.uint8)
    bad = (rng.random(M) < 0.05).astype(np.uint8)
    c = fc.KITTI_CAM
    logs = float(np.log(np.float32(1.2)))
    blob, out = tmp_path / "pin.bin", tmp_path / "pout.bin"
    with open(blob, "wb") as f:
        f.write(struct.pack("<ii", M, nlev))
        f.write(np.asarray(R, np.float32).tobytes() + np.asarray(t, np.float32).tobytes() + np.asarray(Ow, np.float32).tobytes())
        f.write(struct.pack("<4f4f2f", c["fx"], c["fy"], c["cx"], c["cy"], *c["bounds"], c["mbf"], logs))
        for a in (P, Nn, maxd, mind, seen, bad):
            f.write(np.ascontiguousarray(a).tobytes())
    subprocess.check_call([str(exe), str(blob), str(out), "prepass"])
    raw = out.read_bytes()
    n_to_match = struct.unpack_from("<i", raw, 0)[0]
    rec = np.frombuffer(raw, np.dtype([("inview", "<i4"), ("f", "<f4", 5), ("level", "<i4"), ("visible", "<i4"), ("proj", "<i4")]), M, 4)
    F = msorb_mod.Frustum.make(R, t, Ow, c["fx"], c["fy"], c["cx"], c["cy"], c["bounds"], c["mbf"], logs, nlev)
    visit = (seen == 0) & (bad == 0)
    want = oracle.is_in_frustum(F, P[visit], Nn[visit], maxd[visit], mind[visit])
    got = rec[visit]
    assert np.array_equal(got["inview"], want["track_in_view"].astype(np.int32))
    assert got["f"][:, 0].tobytes() == want["proj_x"].tobytes() and got["f"][:, 1].tobytes() == want["proj_y"].tobytes()
    iv = want["track_in_view"] > 0
    assert 200 < iv.sum() < len(iv) and n_to_match == int(iv.sum())
    for col, key in ((2, "proj_xr"), (3, "track_depth"), (4, "view_cos")):
        assert got["f"][iv, col].tobytes() == want[key][iv].tobytes(), key
    assert np.array_equal(got["level"][iv], want["level"][iv])
    assert np.array_equal(got["visible"], iv.astype(np.int32)) and np.array_equal(got["proj"], iv.astype(np.int32))
    untouched = rec[~visit]                                   # mnLastFrameSeen == frame id or isBad(): not visited
    assert np.all(untouched["f"][:, 0] == -7) and np.all(untouched["level"] == -7) and np.all(untouched["visible"] == 0)


def test_cpp_bow_adapters_match_oracle(tmp_path, oracle, msorb_mod):
    """ms-slam_amd/host/BoW_device.h (Frame::ComputeBoW body + the ComputeDistinctiveDescriptors choice) compiled against
    stand-in Frame / DBoW2 container types, vs oracle/bow_oracle.cc."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import bow_cases
    import orb_oracle
    exe = tmp_path / "dropin_bow"
    subprocess.check_call(["g++", "-std=c++17", "-O2", f"-I{ROOT}/tests/cv_stub", f"-I{ROOT}/ms-slam_amd/host",
                           f"-I{ROOT}/include", f"{ROOT}/tests/dropin_bow_main.cc", f"-L{ROOT}/ms-slam_amd", "-lmsorb",
                           f"-Wl,-rpath,{ROOT}/ms-slam_amd", "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib", "-o", str(exe)])
    voc = bow_cases.make_vocabulary(3, k=10, L=4, irregular=True, stop_frac=0.1)
    bow_cases.write_text(tmp_path / "voc.txt", voc)
    feats = bow_cases.make_features(21, voc, 1800)
    obs, ob = bow_cases.make_observations(4, [0, 1, 2, 5, 9, 20, 40, 70] + list(range(3, 30)))
    P = len(ob) - 1
    with open(tmp_path / "in.bin", "wb") as f:
        f.write(struct.pack("<ii", len(feats), P))
        f.write(feats.tobytes())
        f.write(ob.tobytes())
        f.write(obs.tobytes())
    subprocess.check_call([str(exe), str(tmp_path / "voc.txt"), str(tmp_path / "in.bin"), str(tmp_path / "out.bin")])
    blob = (tmp_path / "out.bin").read_bytes()
    nb, nf, same = struct.unpack_from("<iii", blob, 0)
    pos = 12
    words, values = [], []
    for _ in range(nb):
        w, v = struct.unpack_from("<Id", blob, pos)
        pos += 12
        words.append(w)
        values.append(v)
    nodes, lists = [], []
    for _ in range(nf):
        node, cnt = struct.unpack_from("<Ii", blob, pos)
        pos += 8
        lists.append(np.frombuffer(blob, np.uint32, cnt, pos).tolist())
        pos += 4 * cnt
        nodes.append(node)
    best = np.frombuffer(blob, np.int32, P, pos)
    orc = orb_oracle.OracleVocabulary(voc["k"], voc["L"], 0, 0, voc["parent"], voc["is_leaf"], voc["descriptors"],
                                      voc["weights"])
    want = orc.transform(feats, 4)
    assert same == 1 and nb > 100
    assert words == want["bow_word"].tolist()
    assert np.asarray(values, np.float64).tobytes() == want["bow_value"].tobytes()
    assert nodes == want["fv_node"].tolist()
    fb = want["fv_begin"]
    assert lists == [want["fv_feat"][fb[r]:fb[r + 1]].tolist() for r in range(len(nodes))]
    ei, _ = orb_oracle.distinctive_descriptors(obs, ob)
    assert best.tolist() == ei.tolist()


def test_cpp_search_by_bow_adapters_match_oracle(tmp_path, oracle, msorb_mod):
    """SearchByBoWBatch / SearchByBoW / SearchByBoWKeyFrames of ms-slam_amd/host/ORBmatcher_device.h against stand-in
    KeyFrame / Frame / MapPoint types, vs the oracle's restatement of ORBmatcher.cc:223-421 and :872-1016."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import bow_match_cases as bmc
    import orb_oracle
    exe = tmp_path / "dropin_bowmatch"
    subprocess.check_call(["g++", "-std=c++17", "-O2", f"-I{ROOT}/tests/cv_stub", f"-I{ROOT}/ms-slam_amd/host",
                           f"-I{ROOT}/include", f"{ROOT}/tests/dropin_bowmatch_main.cc", f"-L{ROOT}/ms-slam_amd", "-lmsorb",
                           f"-Wl,-rpath,{ROOT}/ms-slam_amd", "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib", "-o", str(exe)])
    rng = np.random.default_rng(5)
    K, ratio, ori = 4, 0.75, 1
    nF = 1500
    descF = rng.integers(0, 256, (nF, 32), dtype=np.uint8)
    nodeF = (rng.integers(0, 40, nF) * 2 + 3).astype(np.int32)
    nodeF[rng.random(nF) < 0.02] = -1
    angF = rng.uniform(0, 360, nF).astype(np.float32)
    sides = [(descF, angF, nodeF, np.zeros(nF, np.uint8))]
    for k in range(K):
        n = 900 + 150 * k
        src = rng.integers(0, nF, n)
        d = bmc.bow_cases._flip_bits(rng, descF[src], rng.integers(0, 30, n))
        node = nodeF[src].copy()
        node[rng.random(n) < 0.1] = 1000
        ang = np.mod(angF[src] + 33.0 + rng.normal(0, 5, n), 360).astype(np.float32)
        ang[rng.random(n) < 0.3] = rng.uniform(0, 360)
        mp = rng.choice([0, 1, 2], n, p=[0.2, 0.7, 0.1]).astype(np.uint8)
        sides.append((np.ascontiguousarray(d), ang, node.astype(np.int32), mp))
    with open(tmp_path / "in.bin", "wb") as f:
        f.write(struct.pack("<iif", K, ori, ratio))
        for d, a, nd, mp in sides:
            f.write(struct.pack("<i", len(d)))
            for arr in (d, a, nd, mp):
                f.write(np.ascontiguousarray(arr).tobytes())
    subprocess.check_call([str(exe), str(tmp_path / "in.bin"), str(tmp_path / "out.bin")])
    res = np.frombuffer((tmp_path / "out.bin").read_bytes(), np.int32)
    fvF = bmc.feature_vector_from_nodes(nodeF)
    pos = 0
    for k in range(K):
        d, a, nd, mp = sides[1 + k]
        nm, m12, m21 = orb_oracle.search_by_bow(d, descF, mp == 1, None, bmc.feature_vector_from_nodes(nd), fvF, a, angF,
                                                50, True, ratio, bool(ori))
        assert res[pos] == nm and nm > 50
        assert res[pos + 1:pos + 1 + nF].tolist() == m21.tolist()
        pos += 1 + nF
    assert res[pos] == 1
    pos += 1
    d0, a0, n0, mp0 = sides[1]
    d1, a1, n1, mp1 = sides[2]
    nm, m12, _ = orb_oracle.search_by_bow(d0, d1, mp0 == 1, mp1 == 1, bmc.feature_vector_from_nodes(n0),
                                          bmc.feature_vector_from_nodes(n1), a0, a1, 50, False, ratio, bool(ori))
    assert res[pos] == nm and nm > 20
    assert res[pos + 1:pos + 1 + len(d0)].tolist() == m12.tolist()


def test_cpp_search_for_triangulation_adapter_matches_oracle(tmp_path, oracle, msorb_mod):
    """SearchForTriangulationBatch / SearchForTriangulation / TriangulationGeometry of ORBmatcher_device.h against
    stand-in KeyFrame / SE3 / camera types (accessor-only KeyFrame, like MS-SLAM's), vs the oracle's restatement of
    ORBmatcher.cc:1168-1402 fed with the F12 / epipole the C++ side computed."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import bow_match_cases as bmc
    import orb_oracle
    exe = tmp_path / "dropin_bowmatch"
    subprocess.check_call(["g++", "-std=c++17", "-O2", f"-I{ROOT}/tests/cv_stub", f"-I{ROOT}/ms-slam_amd/host",
                           f"-I{ROOT}/include", f"{ROOT}/tests/dropin_bowmatch_main.cc", f"-L{ROOT}/ms-slam_amd", "-lmsorb",
                           f"-Wl,-rpath,{ROOT}/ms-slam_amd", "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib", "-o", str(exe)])
    rng = np.random.default_rng(11)
    K = 3
    fx, fy, cx, cy = 718.856, 718.856, 607.19, 185.21
    scale = (np.float32(1.2) ** np.arange(8)).astype(np.float32)
    sigma2 = (scale * scale).astype(np.float32)
    # world points seen by every KeyFrame; KeyFrame poses: small rotations, mostly forward translation
    n_pts = 2500
    Xw = np.stack([rng.uniform(-12, 12, n_pts), rng.uniform(-3, 3, n_pts), rng.uniform(6, 45, n_pts)], 1)
    base_desc = rng.integers(0, 256, (n_pts, 32), dtype=np.uint8)
    base_node = (rng.integers(0, 50, n_pts) * 3 + 1).astype(np.int32)

    def make_kf(seed, shift):
        r = np.random.default_rng(seed)
        a = r.normal(0, 0.02, 3)
        Rz = np.array([[np.cos(a[2]), -np.sin(a[2]), 0], [np.sin(a[2]), np.cos(a[2]), 0], [0, 0, 1]])
        Ry = np.array([[np.cos(a[1]), 0, np.sin(a[1])], [0, 1, 0], [-np.sin(a[1]), 0, np.cos(a[1])]])
        R = (Rz @ Ry).astype(np.float32)
        t = np.array([0.1 * shift, 0.01, -0.7 * shift], np.float32) + r.normal(0, 0.03, 3).astype(np.float32)
        Xc = Xw @ R.T.astype(np.float64) + t
        u = fx * Xc[:, 0] / Xc[:, 2] + cx + r.normal(0, 0.8, n_pts)
        v = fy * Xc[:, 1] / Xc[:, 2] + cy + r.normal(0, 0.8, n_pts)
        vis = np.nonzero((Xc[:, 2] > 1) & (u > 0) & (u < 1241) & (v > 0) & (v < 376) & (r.random(n_pts) < 0.8))[0]
        n = len(vis)
        kp = np.zeros(n, bmc.KP_DTYPE)
        kp["x"], kp["y"] = u[vis], v[vis]
        kp["octave"] = r.integers(0, 8, n)
        kp["angle"] = np.mod(vis * 0.37 + 5.0 * shift + r.normal(0, 4, n), 360)
        desc = bmc.bow_cases._flip_bits(r, base_desc[vis], r.integers(0, 18, n))
        node = base_node[vis].copy()
        node[r.random(n) < 0.05] = -1
        mp = (r.random(n) < 0.35).astype(np.uint8)
        ur = np.where(r.random(n) < 0.5, kp["x"] - r.uniform(1, 30, n), -1).astype(np.float32)
        return dict(desc=np.ascontiguousarray(desc), kp=kp, node=node, mp=mp, ur=ur, R=R, t=t)

    kfs = [make_kf(50, 0)] + [make_kf(51 + k, k + 1) for k in range(K)]
    for only_stereo, coarse, ori in ((0, 0, 1), (1, 0, 1), (0, 1, 0)):
        with open(tmp_path / "tri.bin", "wb") as f:
            f.write(struct.pack("<iiii", K, only_stereo, coarse, ori))
            f.write(struct.pack("<4f", fx, fy, cx, cy))
            for kf in kfs:
                f.write(struct.pack("<i", len(kf["desc"])))
                for arr in (kf["desc"], kf["kp"], kf["node"], kf["mp"], kf["ur"], kf["R"].reshape(9), kf["t"]):
                    f.write(np.ascontiguousarray(arr).tobytes())
                f.write(struct.pack("<i", 8))
                f.write(scale.tobytes())
                f.write(sigma2.tobytes())
        subprocess.check_call([str(exe), str(tmp_path / "tri.bin"), str(tmp_path / "tri_out.bin"), "tri"])
        blob = (tmp_path / "tri_out.bin").read_bytes()
        pos = 0
        a = kfs[0]
        for k in range(K):
            b = kfs[1 + k]
            F12 = np.frombuffer(blob, np.float32, 9, pos)
            ep = np.frombuffer(blob, np.float32, 2, pos + 36)
            nm, cnt = struct.unpack_from("<ii", blob, pos + 44)
            got = np.frombuffer(blob, np.int32, 2 * cnt, pos + 52).reshape(cnt, 2)
            pos += 52 + 8 * cnt
            st1, st2 = a["ur"] >= 0, b["ur"] >= 0
            p = dict(desc1=a["desc"], desc2=b["desc"], valid1=(a["mp"] == 0) & (st1 | (not only_stereo)),
                     avail2=(b["mp"] == 0) & (st2 | (not only_stereo)), stereo1=st1, stereo2=st2,
                     fv1=bmc.feature_vector_from_nodes(a["node"]), fv2=bmc.feature_vector_from_nodes(b["node"]),
                     kp1=a["kp"], kp2=b["kp"], scale_factors2=scale, level_sigma2_2=sigma2, F12=F12, ep=ep)
            wn, w12 = orb_oracle.search_for_triangulation(p, bool(coarse), bool(ori))
            want = np.stack([np.nonzero(w12 >= 0)[0], w12[w12 >= 0]], 1)
            assert nm == wn == cnt and got.tolist() == want.tolist()
            assert nm > 30
            # geometry sanity: the C++ side's epipole is the projection of camera 1's centre into camera 2
            assert np.all(np.isfinite(F12)) and np.all(np.isfinite(ep))
        assert struct.unpack_from("<i", blob, pos)[0] == 1


@pytest.mark.parametrize("motion", ["forward", "backward", "sideways", "mono"])
def test_cpp_search_by_projection_frames_adapter_matches_oracle(tmp_path, oracle, msorb_mod, motion):
    """msorb_host::SearchByProjection(dev, CurrentFrame, LastFrame, th, bMono) (TrackWithMotionModel's matcher call,
    ORBmatcher.cc:1941-2152) over stand-in Frame / MapPoint / SE3 types vs the oracle, which is fed the projections the
    C++ side computed (checked against a float64 projection here)."""
    exe = tmp_path / "dropin_matcher"
    subprocess.check_call(["g++", "-std=c++17", "-O2", f"-I{ROOT}/tests/cv_stub", f"-I{ROOT}/ms-slam_amd/host",
                           f"-I{ROOT}/include", f"{ROOT}/tests/dropin_matcher_main.cc", f"-L{ROOT}/ms-slam_amd", "-lmsorb",
                           f"-Wl,-rpath,{ROOT}/ms-slam_amd", "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib", "-o", str(exe)])
    cfg = synth.KITTI
    ex = msorb_mod.ORBextractor(cfg["nfeatures"], cfg["scale"], cfg["nlevels"], cfg["ini_th"], cfg["min_th"])
    try:
        _, kps, desc = ex(synth.image(6, cfg["rows"], cfg["cols"]))
        scale = np.asarray(ex.GetScaleFactors(), np.float32)
    finally:
        ex.close()
    rng = np.random.default_rng(3)
    N = len(kps)
    fx, fy, cx, cy, mb, mbf = 718.856, 718.856, 607.19, 185.21, 0.54, 386.1448
    bounds = (0.0, float(cfg["cols"]), 0.0, float(cfg["rows"]))
    ur_cur = np.where(rng.random(N) < 0.6, kps["x"] - rng.uniform(1, 40, N), -1).astype(np.float32)
    # current pose = identity-ish; world points = back-projections of the current keypoints (+ noise) so windows hit
    a = 0.01
    Rc = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]], np.float32)
    tc = np.array([0.05, -0.02, 0.1], np.float32)
    NL = 1800
    src = rng.integers(0, N, NL)
    z = rng.uniform(4, 50, NL)
    uu = kps["x"][src] + rng.normal(0, 3, NL)
    vv = kps["y"][src] + rng.normal(0, 3, NL)
    Xc = np.stack([(uu - cx) / fx * z, (vv - cy) / fy * z, z], 1)
    behind = rng.random(NL) < 0.03
    Xc[behind, 2] *= -1
    Xw = ((Xc - tc) @ Rc.astype(np.float64)).astype(np.float32)          # Rc^T (Xc - tc)
    dz = dict(forward=1.5, backward=-1.5, sideways=0.1, mono=1.5)[motion]
    Rl = np.eye(3, dtype=np.float32)
    tl = (tc + np.array([0.0, 0.0, dz], np.float32)).astype(np.float32)   # last camera sits dz behind/ahead along z
    lk = np.zeros(NL, oracle.KP_DTYPE)
    lk["octave"] = np.clip(kps["octave"][src] + rng.integers(-1, 2, NL), 0, 7)
    lk["angle"] = np.mod(kps["angle"][src] + 7.0 + rng.normal(0, 6, NL), 360)
    lk["angle"][rng.random(NL) < 0.2] = rng.uniform(0, 360)
    has = (rng.random(NL) < 0.85).astype(np.uint8)
    outl = (rng.random(NL) < 0.05).astype(np.uint8)
    obs = rng.integers(0, 4, NL).astype(np.int32)
    import bow_cases
    mdesc = bow_cases._flip_bits(rng, desc[src], rng.integers(0, 40, NL))
    held = np.where(rng.random(N) < 0.1, rng.integers(0, 3, N), -1).astype(np.int32)
    th, mono, ori = (15.0, 1, 1) if motion == "mono" else (7.0, 0, 1)
    with open(tmp_path / "fr.bin", "wb") as f:
        f.write(struct.pack("<5i", N, len(scale), NL, mono, ori))
        f.write(struct.pack("<11f", *bounds, fx, fy, cx, cy, mb, mbf, th))
        for arr in (kps, desc, ur_cur, scale, Rc.reshape(9), tc, held, lk, Rl.reshape(9), tl, has, outl, Xw, obs, mdesc):
            f.write(np.ascontiguousarray(arr).tobytes())
    subprocess.check_call([str(exe), str(tmp_path / "fr.bin"), str(tmp_path / "fr_out.bin"), "frames"])
    blob = (tmp_path / "fr_out.bin").read_bytes()
    nm, fwd, bwd = struct.unpack_from("<iii", blob, 0)
    got = np.frombuffer(blob, np.int32, N, 12)
    pos = 12 + 4 * N
    valid = np.frombuffer(blob, np.uint8, NL, pos)
    u = np.frombuffer(blob, np.float32, NL, pos + NL)
    v = np.frombuffer(blob, np.float32, NL, pos + 5 * NL)
    ur = np.frombuffer(blob, np.float32, NL, pos + 9 * NL)
    assert (fwd, bwd) == dict(forward=(1, 0), backward=(0, 1), sideways=(0, 0), mono=(0, 0))[motion]
    # the projections the adapter computed, against float64
    Xc64 = Xw.astype(np.float64) @ Rc.astype(np.float64).T + tc
    u64 = fx * Xc64[:, 0] / Xc64[:, 2] + cx
    v64 = fy * Xc64[:, 1] / Xc64[:, 2] + cy
    want_valid = (has > 0) & (outl == 0) & (Xc64[:, 2] > 0) & (u64 >= 0) & (u64 <= bounds[1]) & (v64 >= 0) & (v64 <= bounds[3])
    edge = (np.abs(u64) < 1e-2) | (np.abs(u64 - bounds[1]) < 1e-2) | (np.abs(v64) < 1e-2) | (np.abs(v64 - bounds[3]) < 1e-2)
    assert np.array_equal(valid[~edge] > 0, want_valid[~edge])
    ok = valid > 0
    assert np.allclose(u[ok], u64[ok], atol=2e-2) and np.allclose(v[ok], v64[ok], atol=2e-2)
    assert np.allclose(ur[ok], u64[ok] - mbf / Xc64[ok, 2], atol=5e-2)
    # oracle on the same projections; ids: last keypoint i -> i, points already held -> NL + k
    rf = oracle.OracleFrame(kps, desc, ur_cur, bounds, scale)
    cur = np.full(N, -1, np.int32)
    hk = np.nonzero(held >= 0)[0]
    cur[hk] = NL + np.arange(len(hk))
    obs_all = np.concatenate([obs, held[hk]]).astype(np.int32)
    last = dict(valid=valid, u=u, v=v, ur=ur, octave=lk["octave"], angle=lk["angle"], desc=mdesc,
                mp=np.arange(NL, dtype=np.int32), obs=obs_all)
    wn = rf.SearchByProjection_frames(last, cur, th, bool(fwd), bool(bwd), bool(ori))
    want = np.where(cur >= NL, -2, cur)                 # the C++ side reports -2 for a point it held before and kept
    assert nm == wn and nm > 100
    assert got.tolist() == want.tolist()
    # the same call with the projection on the device (msorb_host::SearchByProjectionDeviceProjected): the oracle gets the
    # quaternion / translation the C++ side took from the pose and projects with orc_project_last_frame
    subprocess.check_call([str(exe), str(tmp_path / "fr.bin"), str(tmp_path / "frd_out.bin"), "frames_dev"])
    blob = (tmp_path / "frd_out.bin").read_bytes()
    nm1, fwd1, bwd1 = struct.unpack_from("<iii", blob, 0)
    qt = np.frombuffer(blob, np.float32, 7, 12)
    got1 = np.frombuffer(blob, np.int32, N, 40)
    nm2 = struct.unpack_from("<i", blob, 40 + 4 * N)[0]
    got2 = np.frombuffer(blob, np.int32, N, 44 + 4 * N)
    assert (fwd1, bwd1) == (fwd, bwd)
    omm = oracle.MotionModel()
    omm.q[:] = [float(x) for x in qt[:4]]
    omm.t[:] = [float(x) for x in qt[4:]]
    omm.fx, omm.fy, omm.cx, omm.cy, omm.mbf = fx, fy, cx, cy, mbf
    pvalid, pu, pv, pur = oracle.project_last_frame(omm, bounds, ((has > 0) & (outl == 0)).astype(np.uint8), Xw)
    assert np.abs(pu[pvalid > 0] - u64[pvalid > 0]).max() < 2e-2
    for th_k, nm_k, got_k in ((th, nm1, got1), (2 * th, nm2, got2)):
        cur = np.full(N, -1, np.int32)
        cur[hk] = NL + np.arange(len(hk))
        last = dict(valid=pvalid, u=pu, v=pv, ur=pur, octave=lk["octave"], angle=lk["angle"], desc=mdesc,
                    mp=np.arange(NL, dtype=np.int32), obs=obs_all)
        wn = rf.SearchByProjection_frames(last, cur, th_k, bool(fwd), bool(bwd), bool(ori))
        assert nm_k == wn and nm_k > 100
        assert got_k.tolist() == np.where(cur >= NL, -2, cur).tolist()


def test_cpp_fuse_adapter_matches_oracle(tmp_path, oracle, msorb_mod):
    """msorb_host::Fuse (ORBmatcher::Fuse, ORBmatcher.cc:1404-1597) over stand-in KeyFrame / MapPoint types: the device
    search is checked against the oracle's orc_fuse_search on the geometry the C++ side computed, and the mutation log
    (Replace / AddObservation order) against a replay of the reference's loop on the oracle's matches."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import bow_match_cases as bmc
    import matcher_cases as mc
    exe = tmp_path / "dropin_bowmatch"
    subprocess.check_call(["g++", "-std=c++17", "-O2", f"-I{ROOT}/tests/cv_stub", f"-I{ROOT}/ms-slam_amd/host",
                           f"-I{ROOT}/include", f"{ROOT}/tests/dropin_bowmatch_main.cc", f"-L{ROOT}/ms-slam_amd", "-lmsorb",
                           f"-Wl,-rpath,{ROOT}/ms-slam_amd", "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib", "-o", str(exe)])
    cfg = synth.KITTI
    ex = msorb_mod.ORBextractor(cfg["nfeatures"], cfg["scale"], cfg["nlevels"], cfg["ini_th"], cfg["min_th"])
    try:
        _, kps, desc = ex(synth.image(8, cfg["rows"], cfg["cols"]))
        scale = np.asarray(ex.GetScaleFactors(), np.float32)
    finally:
        ex.close()
    rng = np.random.default_rng(21)
    N = len(kps)
    fx, fy, cx, cy, mbf, th = 718.856, 718.856, 607.19, 185.21, 386.1448, 3.0
    logs = float(np.log(np.float32(1.2)))
    sigma2 = (scale * scale).astype(np.float32)
    ur_kf = np.where(rng.random(N) < 0.6, kps["x"] - rng.uniform(1, 40, N), -1).astype(np.float32)
    kf_has = (rng.random(N) < 0.4).astype(np.uint8)
    kf_obs = rng.integers(1, 6, N).astype(np.int32)
    R = np.eye(3, dtype=np.float32)
    t = np.array([0.2, -0.1, 0.3], np.float32)
    M = 2500
    src = rng.integers(0, N, M)
    z = rng.uniform(5, 40, M)
    uu = kps["x"][src] + rng.normal(0, 1.0, M)
    vv = kps["y"][src] + rng.normal(0, 1.0, M)
    Xc = np.stack([(uu - cx) / fx * z, (vv - cy) / fy * z, z], 1)
    Xc[rng.random(M) < 0.03, 2] *= -1
    Xw = (Xc - t).astype(np.float32)                          # R = I
    Ow = (-t).astype(np.float32)
    PO = Xw - Ow
    dist = np.linalg.norm(PO, axis=1)
    normal = (PO / dist[:, None] + rng.normal(0, 0.4, (M, 3))).astype(np.float32)
    normal /= np.linalg.norm(normal, axis=1, keepdims=True)
    lvl = kps["octave"][src]
    maxd = (dist * scale[lvl] * rng.uniform(0.9, 1.1, M)).astype(np.float32)
    mind = (maxd / scale[7] * rng.uniform(0.5, 1.5, M)).astype(np.float32)
    state = rng.choice([0, 1, 2, 3], M, p=[0.03, 0.87, 0.05, 0.05]).astype(np.uint8)
    obs = rng.integers(0, 6, M).astype(np.int32)
    mdesc = mc.flip_bits(rng, desc[src], 30)
    node = np.zeros(N, np.int32)
    with open(tmp_path / "fuse.bin", "wb") as f:
        f.write(struct.pack("<5i", M, 0, cfg["cols"], 0, cfg["rows"]))
        f.write(struct.pack("<7f", fx, fy, cx, cy, mbf, logs, th))
        f.write(struct.pack("<i", N))
        for arr in (desc, kps, node, kf_has, ur_kf, R.reshape(9), t):
            f.write(np.ascontiguousarray(arr).tobytes())
        f.write(struct.pack("<i", 8))
        f.write(scale.tobytes())
        f.write(sigma2.tobytes())
        for arr in (state, Xw, normal, maxd, mind, obs, mdesc, kf_obs):
            f.write(np.ascontiguousarray(arr).tobytes())
    subprocess.check_call([str(exe), str(tmp_path / "fuse.bin"), str(tmp_path / "fuse_out.bin"), "fuse"])
    blob = (tmp_path / "fuse_out.bin").read_bytes()
    n_fused, nlog = struct.unpack_from("<ii", blob, 0)
    log = np.frombuffer(blob, np.int32, nlog, 8).reshape(-1, 3)
    pos = 8 + 4 * nlog
    valid = np.frombuffer(blob, np.uint8, M, pos)
    u, v, ur = (np.frombuffer(blob, np.float32, M, pos + M + 4 * M * k) for k in range(3))
    level = np.frombuffer(blob, np.int32, M, pos + M + 12 * M)
    radius = np.frombuffer(blob, np.float32, M, pos + M + 16 * M)
    assert 1000 < valid.sum() < (state == 1).sum() and np.all(valid[state != 1] == 0)
    # the geometry the C++ side computed, against float64
    ok = valid > 0
    assert np.allclose(u[ok], fx * Xc[ok, 0] / Xc[ok, 2] + cx, atol=2e-2) and np.allclose(v[ok], fy * Xc[ok, 1] / Xc[ok, 2] + cy, atol=2e-2)
    assert np.allclose(radius[ok], th * scale[level[ok]])
    inv_sigma2 = (np.float32(1.0) / sigma2).astype(np.float32)
    rf = oracle.OracleFrame(kps, desc, ur_kf, (0.0, float(cfg["cols"]), 0.0, float(cfg["rows"])), scale)
    bi, bd = rf.FuseSearch(inv_sigma2, valid, u, v, ur, level, radius, mdesc)
    # replay of :1430-1592 with the stand-in map model (Replace: the replaced point turns bad and hands over its
    # observations; AddObservation: +2 observations, the point is now in the KeyFrame)
    pts = {i: dict(obs=int(obs[i]), bad=state[i] == 2, inkf=state[i] == 3) for i in range(M) if state[i]}
    kf_mp = {}
    for j in range(N):
        if kf_has[j]:
            pts[100000 + j] = dict(obs=int(kf_obs[j]), bad=False, inkf=True)
            kf_mp[j] = 100000 + j
    want_log, want_fused = [], 0
    for i in range(M):
        if not state[i] or pts[i]["bad"] or pts[i]["inkf"] or not valid[i]:
            continue
        if bd[i] <= 50:
            j = int(bi[i])
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
    assert n_fused == want_fused and n_fused > 300
    assert log.tolist() == [list(e) for e in want_log]
    kinds = log[:, 0]
    assert (kinds == 1).sum() > 50 and (kinds == 2).sum() > 50
    # the same KeyFrame after map sparsification (KeyFrame::EraseBadDescriptor swapped mGrid away, KeyFrame.cc:355-358):
    # KeyFrame::GetFeaturesInArea returns nothing (:800-801), the reference's Fuse hits `continue` for every point
    # (ORBmatcher.cc:1502-1508) and returns 0 without Replace / AddObservation / AddMapPoint
    subprocess.check_call([str(exe), str(tmp_path / "fuse.bin"), str(tmp_path / "fuse_sp.bin"), "fuse_sparsified"])
    n_fused_sp, nlog_sp = struct.unpack_from("<ii", (tmp_path / "fuse_sp.bin").read_bytes(), 0)
    assert n_fused_sp == 0 and nlog_sp == 0


def test_cpp_relocalisation_search_adapter_matches_oracle(tmp_path, oracle, msorb_mod):
    """msorb_host::SearchByProjection(dev, CurrentFrame, pKF, sAlreadyFound, th, ORBdist) (ORBmatcher.cc:2154-2275) over
    stand-in Frame / KeyFrame / MapPoint types vs the oracle, fed with the projections the C++ side computed."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import matcher_cases as mc
    exe = tmp_path / "dropin_matcher"
    subprocess.check_call(["g++", "-std=c++17", "-O2", f"-I{ROOT}/tests/cv_stub", f"-I{ROOT}/ms-slam_amd/host",
                           f"-I{ROOT}/include", f"{ROOT}/tests/dropin_matcher_main.cc", f"-L{ROOT}/ms-slam_amd", "-lmsorb",
                           f"-Wl,-rpath,{ROOT}/ms-slam_amd", "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib", "-o", str(exe)])
    cfg = synth.KITTI
    ex = msorb_mod.ORBextractor(cfg["nfeatures"], cfg["scale"], cfg["nlevels"], cfg["ini_th"], cfg["min_th"])
    try:
        _, kps, desc = ex(synth.image(9, cfg["rows"], cfg["cols"]))
        scale = np.asarray(ex.GetScaleFactors(), np.float32)
    finally:
        ex.close()
    rng = np.random.default_rng(31)
    N = len(kps)
    fx, fy, cx, cy, th, orb_dist, ori = 718.856, 718.856, 607.19, 185.21, 10.0, 100, 1
    bounds = (0.0, float(cfg["cols"]), 0.0, float(cfg["rows"]))
    logs = float(np.log(np.float32(1.2)))
    ur_cur = np.where(rng.random(N) < 0.6, kps["x"] - rng.uniform(1, 40, N), -1).astype(np.float32)
    Rc = np.eye(3, dtype=np.float32)
    tc = np.array([0.1, 0.05, -0.2], np.float32)
    n = 1700
    src = rng.integers(0, N, n)
    z = rng.uniform(4, 50, n)
    uu = kps["x"][src] + rng.normal(0, 4, n)
    vv = kps["y"][src] + rng.normal(0, 4, n)
    Xc = np.stack([(uu - cx) / fx * z, (vv - cy) / fy * z, z], 1)
    Xw = (Xc - tc).astype(np.float32)
    dist = np.linalg.norm(Xw - (-tc), axis=1)
    lvl = kps["octave"][src]
    maxd = (dist * scale[lvl] * rng.uniform(0.85, 1.15, n)).astype(np.float32)
    mind = (maxd / scale[7] * rng.uniform(0.5, 1.4, n)).astype(np.float32)
    kk = np.zeros(n, oracle.KP_DTYPE)
    kk["angle"] = np.mod(kps["angle"][src] + 20.0 + rng.normal(0, 6, n), 360)
    kk["angle"][rng.random(n) < 0.2] = rng.uniform(0, 360)
    state = rng.choice([0, 1, 2, 3], n, p=[0.1, 0.75, 0.05, 0.1]).astype(np.uint8)
    mdesc = mc.flip_bits(rng, desc[src], 40)
    held = (rng.random(N) < 0.15).astype(np.uint8)
    with open(tmp_path / "rl.bin", "wb") as f:
        f.write(struct.pack("<5i", N, len(scale), n, orb_dist, ori))
        f.write(struct.pack("<10f", *bounds, fx, fy, cx, cy, logs, th))
        for arr in (kps, desc, ur_cur, scale, Rc.reshape(9), tc, held, kk, state, Xw, maxd, mind, mdesc):
            f.write(np.ascontiguousarray(arr).tobytes())
    subprocess.check_call([str(exe), str(tmp_path / "rl.bin"), str(tmp_path / "rl_out.bin"), "reloc"])
    blob = (tmp_path / "rl_out.bin").read_bytes()
    nm = struct.unpack_from("<i", blob, 0)[0]
    got = np.frombuffer(blob, np.int32, N, 4)
    pos = 4 + 4 * N
    valid = np.frombuffer(blob, np.uint8, n, pos)
    u = np.frombuffer(blob, np.float32, n, pos + n)
    v = np.frombuffer(blob, np.float32, n, pos + 5 * n)
    level = np.frombuffer(blob, np.int32, n, pos + 9 * n)
    assert 500 < valid.sum() < (state == 1).sum() and np.all(valid[state != 1] == 0)
    ok = valid > 0
    assert np.allclose(u[ok], fx * Xc[ok, 0] / Xc[ok, 2] + cx, atol=2e-2) and np.allclose(v[ok], fy * Xc[ok, 1] / Xc[ok, 2] + cy, atol=2e-2)
    rf = oracle.OracleFrame(kps, desc, ur_cur, bounds, scale)
    cur = np.where(held > 0, n + np.arange(N), -1).astype(np.int32)
    pts = dict(valid=valid, u=u, v=v, level=level, angle=kk["angle"], desc=mdesc, mp=np.arange(n, dtype=np.int32))
    wn = rf.SearchByProjection_kf(pts, cur, th, orb_dist, bool(ori))
    want = np.where(cur >= n, -2, cur)
    assert nm == wn and nm > 100
    assert got.tolist() == want.tolist()


def test_cpp_stereo_frame_constructor_one_call(tmp_path, oracle, msorb_mod):
    """msorb_host::ExtractStereo (the two ExtractORB threads + ComputeStereoMatches of Frame.cc:119-137 as one device call)
    fills the Frame exactly like the reference's sequence through the drop-in class, and like the oracle."""
    import matcher_cases as mc
    exe = tmp_path / "dropin_extractor"
    subprocess.check_call(["g++", "-std=c++17", "-O2", f"-I{ROOT}/tests/cv_stub", f"-I{ROOT}/ms-slam_amd/host",
                           f"-I{ROOT}/include", f"{ROOT}/tests/dropin_extractor_main.cc",
                           f"{ROOT}/ms-slam_amd/host/ORBextractor.cc", f"-L{ROOT}/ms-slam_amd", "-lmsorb",
                           f"-Wl,-rpath,{ROOT}/ms-slam_amd", "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib", "-o", str(exe)])
    cfg = synth.KITTI
    L, R = synth.stereo_pair(55, cfg["rows"], cfg["cols"])
    L.tofile(tmp_path / "l.raw")
    R.tofile(tmp_path / "r.raw")
    mbf, mb = mc.KITTI_BF, mc.KITTI_BF / mc.KITTI_FX
    n_dev = msorb_mod.lib().msorb_device_count()
    # five extractor objects are built before the split pair (exL, exR, exF, then exS0, exS1): the device list is written
    # so that exS0 / exS1 land on devices 0 / 1 when the box has two GPUs
    env = dict(os.environ, MSORB_DEVICES="0,0,0,0,1" if n_dev >= 2 else "0")
    subprocess.check_call([str(exe), str(cfg["rows"]), str(cfg["cols"]), str(tmp_path / "l.raw"), str(tmp_path / "out.bin"), "2000",
                           "stereo", str(tmp_path / "r.raw"), repr(float(np.float32(mb))), repr(float(np.float32(mbf)))], env=env)
    blob = (tmp_path / "out.bin").read_bytes()
    devs = struct.unpack_from("<ii", blob, 0)
    assert devs == ((0, 1) if n_dev >= 2 else (0, 0))      # construction order -> device (ORBextractor.cc next_device)
    pos, frames = 8, []
    for _ in range(5):
        n, nr = struct.unpack_from("<ii", blob, pos)
        pos += 8
        kl = np.frombuffer(blob, oracle.KP_DTYPE, n, pos); pos += 28 * n
        kr = np.frombuffer(blob, oracle.KP_DTYPE, nr, pos); pos += 28 * nr
        dl = np.frombuffer(blob, np.uint8, 32 * n, pos).reshape(n, 32); pos += 32 * n
        dr = np.frombuffer(blob, np.uint8, 32 * nr, pos).reshape(nr, 32); pos += 32 * nr
        ur = np.frombuffer(blob, np.float32, n, pos); pos += 4 * n
        dp = np.frombuffer(blob, np.float32, n, pos); pos += 4 * n
        frames.append((kl, kr, dl, dr, ur, dp))
    a, b, c, d, e = frames  # ExtractStereo, the reference's sequence, ExtractStereoSplit (one object per device) twice, ExtractStereoFrame
    for other in (b, c, d, e):
        for x, y in zip(a, other):
            assert x.tobytes() == y.tobytes()
    _, okl, odl = oracle.OracleExtractor(2000, 1.2, 8, 20, 7)(L)
    assert np.array_equal(a[0].view(np.uint8), okl.view(np.uint8)) and np.array_equal(a[2], odl)
    assert (a[4] > 0).sum() > 500
    # the device frame ExtractStereoFrame left behind: its grid is Frame::AssignFeaturesToGrid of the oracle
    n_assigned = struct.unpack_from("<i", blob, pos)[0]
    pos += 4
    cb = np.frombuffer(blob, np.int32, 64 * 48 + 1, pos); pos += 4 * (64 * 48 + 1)
    ci = np.frombuffer(blob, np.int32, n_assigned, pos)
    rf = oracle.OracleFrame(e[0], e[2], e[4], (0.0, float(cfg["cols"]), 0.0, float(cfg["rows"])), (np.float32(1.2) ** np.arange(8)).astype(np.float32))
    rcb, rci = rf.grid_csr()
    assert np.array_equal(cb, rcb) and np.array_equal(ci, rci) and n_assigned == len(e[0])


def test_cpp_search_local_points_chain_matches_oracle(tmp_path, oracle, msorb_mod):
    """msorb_host::SearchLocalPoints (Tracking::SearchLocalPoints from its second loop on — the isInFrustum loop AND the
    SearchByProjection call — as one device chain) over stand-in Frame / MapPoint objects vs the oracle's composition of
    Frame::isInFrustum and ORBmatcher::SearchByProjection: nmatches, nToMatch, F.mvpMapPoints, mbTrackInView, IncreaseVisible,
    mmProjectPoints."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import frustum_cases as fc
    import matcher_cases as mc
    import track_cases as tc
    exe = tmp_path / "dropin_matcher"
    subprocess.check_call(["g++", "-std=c++17", "-O2", f"-I{ROOT}/tests/cv_stub", f"-I{ROOT}/ms-slam_amd/host",
                           f"-I{ROOT}/include", f"{ROOT}/tests/dropin_matcher_main.cc", f"-L{ROOT}/ms-slam_amd", "-lmsorb",
                           f"-Wl,-rpath,{ROOT}/ms-slam_amd", "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib", "-o", str(exe)])
    cfg = synth.KITTI
    L, Rimg = synth.stereo_pair(33, cfg["rows"], cfg["cols"])
    ex = msorb_mod.ORBextractor(2000, 1.2, 8, 20, 7)
    mbf, mb = mc.KITTI_BF, mc.KITTI_BF / mc.KITTI_FX
    try:
        kl, dl, kr, dr, ur, dp, _ = ex.extract_stereo(L, Rimg, mb, mbf)
        scale = np.asarray(ex.GetScaleFactors(), np.float32)
    finally:
        ex.close()
    N, M, nlev = len(kl), 5000, 8
    rng = np.random.default_rng(6)
    w = rng.normal(scale=0.02, size=3)
    ang = np.linalg.norm(w)
    k = w / ang
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    R = (np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * K @ K).astype(np.float32)
    t = rng.normal(scale=0.3, size=3).astype(np.float32)
    Ow = (-(R.T.astype(np.float64) @ t.astype(np.float64))).astype(np.float32)
    mp = tc.local_map(14, kl, dl, ur, dp, R, t, Ow, scale, M)
    seen = (1 - mp["visit"]).astype(np.uint8)      # "already matched in this frame": mnLastFrameSeen == mnId
    c = fc.KITTI_CAM
    logs = float(np.log(np.float32(1.2)))
    th, far, th_far, nnratio = 3.0, 1, 45.0, 0.8
    init = np.where(rng.random(N) < 0.15, rng.integers(0, M, N), -1).astype(np.int32)
    blob, out = tmp_path / "lin.bin", tmp_path / "lout.bin"
    with open(blob, "wb") as f:
        f.write(struct.pack("<4i", N, nlev, M, far))
        f.write(np.asarray(R, np.float32).tobytes() + np.asarray(t, np.float32).tobytes() + np.asarray(Ow, np.float32).tobytes())
        f.write(struct.pack("<4f4f2f3f", c["fx"], c["fy"], c["cx"], c["cy"], *c["bounds"], c["mbf"], logs, th, th_far, nnratio))
        for a in (kl, dl, ur, scale, mp["pos_w"], mp["normal"], mp["max_distance"], mp["min_distance"], seen, mp["bad"], mp["sparsified"],
                  mp["obs"], mp["desc"], init):
            f.write(np.ascontiguousarray(a).tobytes())
    subprocess.check_call([str(exe), str(blob), str(out), "local"])
    raw = out.read_bytes()
    nmatches, n_to_match = struct.unpack_from("<2i", raw, 0)
    got_mp = np.frombuffer(raw, np.int32, N, 8)
    rec = np.frombuffer(raw, np.dtype([("inview", "<i4"), ("visible", "<i4"), ("proj", "<i4")]), M, 8 + 4 * N)
    # the oracle's composition.  The C++ side hands the matcher only part of the table as "local" points (held points with an
    # odd index are frame-held extras): those are never queries
    held = np.zeros(M, bool)
    held[init[init >= 0]] = True
    not_local = held & (np.arange(M) % 2 == 1)
    mpo = dict(mp)
    mpo["visit"] = (mp["visit"].astype(bool) & ~mp["bad"].astype(bool) & ~not_local).astype(np.uint8)
    F = msorb_mod.Frustum.make(R, t, Ow, c["fx"], c["fy"], c["cx"], c["cy"], c["bounds"], c["mbf"], logs, nlev)
    rf = oracle.OracleFrame(kl, dl, ur, c["bounds"], scale)
    want_mp = init.copy()
    rnm, r, visit = tc.oracle_local_points(oracle, rf, F, mpo, want_mp, th, bool(far), th_far, nnratio)
    assert nmatches == rnm > 300
    assert np.array_equal(got_mp, want_mp)
    iv = r["track_in_view"].astype(bool) & visit
    assert n_to_match == int(iv.sum())
    assert np.array_equal(rec["inview"].astype(bool), iv)
    assert np.array_equal(rec["visible"], iv.astype(np.int32)) and np.array_equal(rec["proj"], iv.astype(np.int32))

