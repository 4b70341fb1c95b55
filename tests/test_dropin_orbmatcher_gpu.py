"""The drop-in class ORB_SLAM3::ORBmatcher (ms-slam_amd/host/ORBmatcher.{h,cc} = the declaration of the reference's
include/ORBmatcher.h:36-112) compiled against the stand-ins of tests/slam_stub, linked with libmsorb.so and driven like
LoopClosing.cc / Tracking.cc drive the reference: SearchBySim3, the three SearchByProjection(pKF, Scw, ...) forms,
Fuse(pKF, Scw, ...), SearchForInitialization, the loop form of SearchByBoW, DescriptorDistance.  Every result is compared
with the oracle, fed with the projections the C++ side computed (the Sim3 / SE3 arithmetic is the caller's code)."""
import os
import struct
import subprocess
import sys

import numpy as np
import pytest

from msorb import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
pytestmark = pytest.mark.gpu


class Reader:
    def __init__(self, blob):
        self.b, self.p = blob, 0

    def arr(self, dt, n):
        a = np.frombuffer(self.b, dt, n, self.p)
        self.p += a.nbytes
        return a

    def i(self):
        return int(self.arr(np.int32, 1)[0])

    def queries(self, n):
        return dict(valid=self.arr(np.uint8, n), u=self.arr(np.float32, n), v=self.arr(np.float32, n), level=self.arr(np.int32, n))


def test_orbmatcher_class_loop_closing_and_initialisation(tmp_path, oracle, msorb_mod):
    import matcher_cases as mc
    import bow_match_cases as bmc
    exe = tmp_path / "dropin_orbmatcher"
    subprocess.check_call(["g++", "-std=c++17", "-O2", f"-I{ROOT}/tests/slam_stub", f"-I{ROOT}/tests/cv_stub",
                           f"-I{ROOT}/ms-slam_amd/host", f"-I{ROOT}/include", f"{ROOT}/tests/dropin_orbmatcher_main.cc",
                           f"{ROOT}/ms-slam_amd/host/ORBmatcher.cc", f"-L{ROOT}/ms-slam_amd", "-lmsorb",
                           f"-Wl,-rpath,{ROOT}/ms-slam_amd", "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib", "-lpthread",
                           "-o", str(exe)])
    cfg = synth.KITTI
    rows, cols = cfg["rows"], cfg["cols"]
    rng = np.random.Generator(np.random.PCG64(77))
    A = synth.image(21, rows, cols)
    B = np.clip(np.roll(A, (2, 5), (0, 1)).astype(np.int32) + rng.integers(-3, 4, A.shape), 0, 255).astype(np.uint8)
    ex = msorb_mod.ORBextractor(2000, 1.2, 8, 20, 7)
    try:
        _, k1, d1 = ex(A)
        _, k2, d2 = ex(B)
        scale = np.asarray(ex.GetScaleFactors(), np.float32)
    finally:
        ex.close()
    sigma2 = (scale * scale).astype(np.float32)
    N1, N2, P = len(k1), len(k2), 3000
    fx, fy, cx, cy = 718.856, 718.856, 607.19, 185.2
    logs = float(np.log(np.float32(1.2)))
    z = 10.0
    t2 = np.array([5 * z / fx, 2 * z / fy, 0.0], np.float32)
    th_sim3, th_proj, ratio = 7.5, 8, 1.5

    def points(k, d, shift_w, n_state):
        n = len(k)
        pos = np.stack([(k["x"] - cx) / fx * z, (k["y"] - cy) / fy * z, np.full(n, z)], 1) + shift_w
        pos = (pos + rng.normal(0, 0.004, pos.shape)).astype(np.float32)
        dist = np.linalg.norm(pos, axis=1)
        nrm = (pos / dist[:, None]).astype(np.float32)
        flip = rng.random(n) < 0.05
        nrm[flip] *= -1                                               # viewed from behind: fails the 60-degree test
        maxd = (dist * scale[k["octave"]] * rng.uniform(0.9, 1.0, n)).astype(np.float32)
        mind = (maxd / scale[7] * 0.8).astype(np.float32)
        state = rng.choice([0, 1, 2], n, p=n_state).astype(np.uint8)
        desc = mc.flip_bits(rng, d, 25)
        return state, pos, nrm, maxd, mind, desc

    s1 = points(k1, d1, 0.0, [0.15, 0.8, 0.05])
    s2 = points(k2, d2, -t2, [0.15, 0.8, 0.05])                       # KF2's points: world = camera-2 coordinates - t2
    node1 = ((k1["x"] // 64).astype(np.int32) + 20 * (k1["y"] // 64).astype(np.int32)).astype(np.int32)
    node2 = (((k2["x"] - 5) // 64).astype(np.int32) + 20 * ((k2["y"] - 2) // 64).astype(np.int32)).astype(np.int32)
    node2 = np.maximum(node2, 0)
    already12 = np.full(N1, -1, np.int32)
    pick = rng.choice(np.nonzero(s1[0] == 1)[0], 40, replace=False)
    already12[pick] = rng.choice(np.nonzero(s2[0] == 1)[0], 40, replace=False)
    src = rng.integers(0, N1, P)
    cand = points(k1[src], d1[src], 0.0, [0.05, 0.85, 0.10])
    matched_init = np.full(N2, -1, np.int32)
    good_c = np.nonzero(cand[0] == 1)[0]
    slots = rng.choice(N2, 60, replace=False)
    matched_init[slots] = rng.choice(good_c, 60, replace=False)
    prev = (np.stack([k1["x"], k1["y"]], 1) + rng.normal(0, 2.0, (N1, 2))).astype(np.float32)
    loop1 = (rng.random(N1) < 0.1).astype(np.uint8)
    loop2 = (rng.random(N2) < 0.1).astype(np.uint8)
    with open(tmp_path / "in.bin", "wb") as f:
        f.write(struct.pack("<6i", N1, N2, 8, P, rows, cols))
        f.write(struct.pack("<8f", fx, fy, cx, cy, logs, th_sim3, th_proj, ratio))
        f.write(scale.tobytes()); f.write(sigma2.tobytes())
        for k, d, s, node in ((k1, d1, s1, node1), (k2, d2, s2, node2)):
            f.write(np.ascontiguousarray(k).tobytes()); f.write(np.ascontiguousarray(d).tobytes())
            for a in s:
                f.write(np.ascontiguousarray(a).tobytes())
            f.write(node.tobytes())
        f.write(t2.tobytes()); f.write(already12.tobytes())
        for a in cand:
            f.write(np.ascontiguousarray(a).tobytes())
        f.write(matched_init.tobytes()); f.write(prev.tobytes()); f.write(loop1.tobytes()); f.write(loop2.tobytes())
    subprocess.check_call([str(exe), str(tmp_path / "in.bin"), str(tmp_path / "out.bin")])
    R = Reader((tmp_path / "out.bin").read_bytes())
    bounds = (0.0, float(cols), 0.0, float(rows))
    r1 = oracle.OracleFrame(k1, d1, None, bounds, scale)
    r2 = oracle.OracleFrame(k2, d2, None, bounds, scale)
    fxf, cxf = np.float32(fx), np.float32(cx)

    # ---- SearchBySim3
    nFound = R.i()
    ids = R.arr(np.int32, N1)
    q1, q2 = R.queries(N1), R.queries(N2)
    q1["desc"], q2["desc"] = s1[5], s2[5]
    assert 800 < q1["valid"].sum() < (s1[0] == 1).sum() and np.all(q1["valid"][s1[0] != 1] == 0)
    ok = q1["valid"] > 0
    assert np.allclose(q1["u"][ok], fxf * (s1[1][ok, 0] + t2[0]) / s1[1][ok, 2] + cxf, atol=2e-2)   # geometry vs float64
    assert np.all(q1["valid"][already12 >= 0] == 0) and np.all(q2["valid"][already12[already12 >= 0]] == 0)
    want12, wf = oracle.search_by_sim3(r1, r2, q1, q2, th_sim3)
    want_ids = np.where(already12 >= 0, already12, want12)
    assert nFound == wf and nFound > 300 and np.array_equal(ids, want_ids)

    # ---- SearchByProjection(pKF, Scw, vpPoints, vpMatched, th, ratioHamming)
    n_a = R.i()
    ids_a = R.arr(np.int32, N2)
    qa = R.queries(P)
    qa["desc"] = cand[5]
    qa["mp"] = np.arange(P, dtype=np.int32)
    assert np.all(qa["valid"][cand[0] != 1] == 0) and np.all(qa["valid"][matched_init[matched_init >= 0]] == 0)
    matched = np.where(matched_init >= 0, P + np.arange(N2), -1).astype(np.int32)
    wn = r2.SearchByProjection_sim3(qa, matched, float(th_proj), float(np.float32(50) * np.float32(ratio)))
    want_a = np.where(matched >= P, matched_init, matched)
    assert n_a == wn and n_a > 500 and np.array_equal(ids_a, want_a)

    # ---- the (pKF, Scw, vpPoints, vpPointsKFs, vpMatched, vpMatchedKF, ...) form: only existing candidates are passed
    n_b = R.i()
    G = R.i()
    good_idx = R.arr(np.int32, G)
    ids_b = R.arr(np.int32, N2)
    kf_b = R.arr(np.int32, N2)
    qb = R.queries(G)
    assert np.array_equal(good_idx, np.nonzero(cand[0] != 0)[0])
    qb["desc"] = cand[5][good_idx]
    qb["mp"] = np.arange(G, dtype=np.int32)
    matched = np.where(matched_init >= 0, G + np.arange(N2), -1).astype(np.int32)
    wn = r2.SearchByProjection_sim3(qb, matched, float(th_proj), float(np.float32(50) * np.float32(ratio)))
    want_b = np.where(matched >= G, matched_init, np.where(matched >= 0, good_idx[np.maximum(matched, 0) % G], -1))
    assert n_b == wn and np.array_equal(ids_b, want_b)
    new = (matched >= 0) & (matched < G)
    assert np.array_equal(kf_b[new], np.where(matched[new] & 1, 1, 2)) and np.all(kf_b[~new] == -1)
    # hand projection (fx * (X * invz) + cx) and mpCamera->project (fx * X / Z + cx) differ in the last bit for some points
    assert np.allclose(qb["u"], qa["u"][good_idx], atol=1e-3)

    # ---- SearchByProjectionLoop
    n_l = R.i()
    res = R.arr(np.int32, G)
    okkf = R.i()
    ql = R.queries(G)
    ql["desc"] = cand[5][good_idx]
    train_ok = (s2[0] == 1).astype(np.uint8)
    wi, wn = oracle.search_by_projection_loop(r2, ql, train_ok, float(th_proj), float(np.float32(50) * np.float32(ratio)))
    pre = np.zeros(G, bool)
    pre[::7] = True
    assert np.all(ql["valid"][pre] == 0)
    assert n_l == wn and n_l > 300 and okkf == 1
    assert np.array_equal(res, np.where(pre, -3, wi))

    # ---- Fuse(pKF, Scw, vpPoints, th, vpReplacePoint)
    nFused = R.i()
    repl = R.arr(np.int32, G)
    nlog = R.i()
    log = R.arr(np.int32, nlog).reshape(-1, 3)
    mp_ids = R.arr(np.int32, G)
    qf = R.queries(G)
    qf["desc"] = cand[5][good_idx]
    bi, bd = oracle.fuse_sim3_search(r2, qf, 4.0)
    slot = {j: ("kf", j) for j in range(N2) if s2[0][j] != 0}            # GetMapPoint(j)
    want_repl, want_log, want_fused = np.full(G, -1, np.int32), [], 0
    for i in range(G):
        if not qf["valid"][i] or bd[i] > 50:
            continue
        j = int(bi[i])
        if j in slot:
            kind, x = slot[j]
            if kind == "cand":
                want_repl[i] = -2                                    # a candidate added earlier in this very loop (:1707-1711)
            elif s2[0][x] == 1:
                want_repl[i] = x
        else:
            want_log.append((2, int(mp_ids[i]), j))
            slot[j] = ("cand", i)
        want_fused += 1
    assert nFused == want_fused and nFused > 200
    assert np.array_equal(repl, want_repl) and log.tolist() == [list(e) for e in want_log] and len(want_log) > 20

    # ---- SearchForInitialization
    n_i = R.i()
    m12 = R.arr(np.int32, N1)
    prev_out = R.arr(np.float32, 2 * N1).reshape(N1, 2)
    want_prev = prev.copy()
    w12, wn = oracle.search_for_initialization(r1, r2, want_prev, 100, 0.9, True)
    assert n_i == wn and n_i > 50 and np.array_equal(m12, w12)
    assert np.array_equal(prev_out.view(np.uint32), want_prev.view(np.uint32))

    # ---- SearchByBoW, loop form: the KeyFrame-KeyFrame search with mnLoopPointForKF exclusions, output in histogram-bin order
    n_w = R.i()
    cmp_ids = R.arr(np.int32, n_w)
    lmp_ids = R.arr(np.int32, n_w)
    consistent = R.i()
    # kf2 holds the candidates Fuse added: they carry no loop mark and are good
    added = {j for j, (kind, _) in slot.items() if kind == "cand"}
    v1 = ((s1[0] == 1) & (loop1 == 0)).astype(np.uint8)
    a2 = np.array([(j in added) or (s2[0][j] == 1 and loop2[j] == 0) for j in range(N2)], np.uint8)
    fv1, fv2 = bmc.feature_vector_from_nodes(node1), bmc.feature_vector_from_nodes(node2)
    nm, w12, _ = oracle.search_by_bow(d1, d2, v1, a2, fv1, fv2, k1["angle"], k2["angle"], 50, False, 0.9, True)
    order = []
    hist = [[] for _ in range(30)]
    for node, b, e in zip(fv1[0], fv1[1][:-1], fv1[1][1:]):
        if node not in set(fv2[0].tolist()):
            continue
        for idx1 in fv1[2][b:e]:
            if w12[idx1] >= 0:
                rot = np.float32(k1["angle"][idx1]) - np.float32(k2["angle"][w12[idx1]])
                if rot < 0:
                    rot = np.float32(rot + np.float32(360.0))
                bn = int(np.round(np.float32(rot * np.float32(1.0 / 30))))     # round-half-even == C round() away from ties? ties absent in float data
                hist[0 if bn == 30 else bn].append(int(idx1))
    for h in hist:
        order += h
    assert n_w == nm == len(order) and n_w > 100 and consistent == 1
    assert cmp_ids.tolist() == order
    assert lmp_ids.tolist() == [int(w12[i]) for i in order]      # (ids in kf2's CURRENT map-point vector: Fuse's additions included)

    # ---- SearchForTriangulation / SearchByBoW(pKF, F): the class (resident KeyFrame store) equals the per-call device path
    n_tri, n_bow, same = R.i(), R.i(), R.i()
    assert same == 1 and n_bow > 100 and n_tri >= 0

    # ---- DescriptorDistance, constants
    acc = R.i()
    consts = R.i()
    n = min(N1, N2)
    want_acc = int(np.unpackbits(np.bitwise_xor(d1[:n], d2[:n]), axis=1).sum())
    assert acc == want_acc and consts == 50 * 10000 + 100 * 100 + 30
    assert R.p == len(R.b)
