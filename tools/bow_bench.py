#!/usr/bin/env python3
"""Throughput of the two "next" rows of SURVEY.md §8f built after the front-end:
  * DBoW2 transform (Frame::ComputeBoW) on msorb_extract_batch's device descriptors, ORBvoc-shaped synthetic
    vocabulary (k=10, L=6, 1 111 111 nodes / 1 000 000 words; the real ORBvoc.txt is not in the checkout),
  * MapPoint::ComputeDistinctiveDescriptors batched over a synthetic map.
Each with the oracle timed on a bounded sample on this box's host cores.  Prints one JSON object."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "ms-slam_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import torch
import msorb
from msorb import synth
import bow_cases
import orb_oracle


def main(pairs=128, L=6):
    cfg = synth.KITTI
    res = {}
    t0 = time.perf_counter()
    voc = bow_cases.make_vocabulary(0, k=10, L=L, stop_frac=0.01)
    res["vocabulary"] = dict(k=10, L=L, nodes=int(len(voc["parent"])), words=int(voc["is_leaf"].sum()),
                             device_bytes=int((len(voc["parent"]) - 1) * (32 + 16 + 8)),
                             build_s=round(time.perf_counter() - t0, 1))
    dev = msorb.Vocabulary(voc["k"], voc["L"], 0, 0, voc["parent"], voc["is_leaf"], voc["descriptors"], voc["weights"])
    ex = msorb.ORBextractor(cfg["nfeatures"], cfg["scale"], cfg["nlevels"], cfg["ini_th"], cfg["min_th"])
    base = synth.stereo_batch(8, cfg["rows"], cfg["cols"], seed0=0)
    host = np.concatenate([base] * (pairs // 8 + 1))[:2 * pairs]
    d_img = torch.from_numpy(np.ascontiguousarray(host)).cuda()
    counts, _, d_kps, d_desc = ex.extract_batch(d_img, (0, 0))
    # real extractor descriptors are far from the synthetic clusters; overwrite 3/4 of them with noisy leaves so that the
    # descents spread over the whole tree the way real words do (the timing depends on the memory pattern only)
    n_tot = int(counts.sum())
    feats = bow_cases.make_features(1, voc, 4096)
    rep = torch.from_numpy(feats).cuda()
    idx = torch.randint(0, 4096, (d_desc.shape[0], d_desc.shape[1]), device="cuda")
    mask = (torch.rand(d_desc.shape[:2], device="cuda") < 0.75)
    d_desc[mask] = rep[idx[mask]]
    for _ in range(3):
        out = dev.transform_batch(d_desc, counts)
    ms = [dev.transform_batch(d_desc, counts)["elapsed_ms"] for _ in range(10)]
    t0 = time.perf_counter()
    for _ in range(5):
        dev.transform_batch(d_desc, counts)
    wall = (time.perf_counter() - t0) / 5
    m = float(np.median(ms))
    levels = L
    res["bow_transform"] = dict(frames=int(len(counts)), descriptors=n_tot, kernel_ms=round(m, 4), wall_ms=round(wall * 1e3, 4),
                                mdescriptors_per_s=round(n_tot / m / 1e3, 2),
                                ghamming_per_s=round(n_tot * 10 * levels / m / 1e6, 2),
                                words_per_frame=float(out["n_bow"].float().mean()),
                                nodes_per_frame=float(out["n_fv"].float().mean()))
    # CPU oracle on a sample of frames
    orc = orb_oracle.OracleVocabulary(voc["k"], voc["L"], 0, 0, voc["parent"], voc["is_leaf"], voc["descriptors"], voc["weights"])
    NS = 64
    dh = d_desc[:NS].cpu().numpy()
    t0 = time.perf_counter()
    nd = 0
    for i in range(NS):
        r = orc.transform(dh[i, :counts[i]])
        nd += int(counts[i])
    dt = time.perf_counter() - t0
    # parity of the sample while we are here
    for i in range(8):
        r = orc.transform(dh[i, :counts[i]])
        nb = int(out["n_bow"][i])
        assert out["bow_word"][i, :nb].cpu().numpy().tolist() == r["bow_word"].tolist()
        assert out["bow_value"][i, :nb].cpu().numpy().tobytes() == r["bow_value"].tobytes()
    res["bow_transform"]["cpu_oracle_mdescriptors_per_s"] = round(nd / dt / 1e6, 4)
    res["bow_transform"]["cpu_sample"] = f"{NS} frames, {nd} descriptors, 1 thread, {dt:.2f} s"
    dev.close(); ex.close()

    # distinctive descriptors: 200k map points, observations ~ 2 + Geom(mean 8), a few long tracks
    rng = np.random.default_rng(0)
    sizes = 2 + rng.geometric(1 / 8.0, 200000)
    sizes[rng.integers(0, len(sizes), 200)] = rng.integers(65, 300, 200)
    desc, ob = bow_cases.make_observations(5, sizes[:2000])
    reps = len(sizes) // 2000
    desc = np.tile(desc, (reps, 1))
    ob = np.concatenate([[0], np.cumsum(np.tile(np.diff(ob), reps))]).astype(np.int32)
    pairs_n = int((np.diff(ob).astype(np.int64) ** 2).sum())
    msorb.distinctive_descriptors(desc, ob)
    t0 = time.perf_counter()
    bi, bm, kms = msorb.distinctive_descriptors(desc, ob)
    wall = time.perf_counter() - t0
    t0 = time.perf_counter()
    ei, em = orb_oracle.distinctive_descriptors(desc[:ob[20000]], ob[:20001])
    dt = time.perf_counter() - t0
    assert bi[:20000].tolist() == ei.tolist() and bm[:20000].tolist() == em.tolist()
    res["distinctive_descriptors"] = dict(points=int(len(ob) - 1), descriptors=int(ob[-1]), matrix_entries=pairs_n,
                                          kernel_ms=round(kms, 4), wall_ms=round(wall * 1e3, 3),
                                          mpoints_per_s=round((len(ob) - 1) / kms / 1e3, 2),
                                          gpairs_per_s=round(pairs_n / kms / 1e6, 2),
                                          cpu_oracle_mpoints_per_s=round(20000 / dt / 1e6, 4),
                                          cpu_sample=f"20000 points, 1 thread, {dt:.2f} s")
    # SearchByBoW: one reference-frame pair (TrackReferenceKeyFrame) and a relocalisation batch of 32 candidates,
    # 2000 features a side, 100 level-2 nodes (levelsup 4 of L 6)
    import bow_match_cases as bmc
    cand = [bmc.make_pair(40 + i, n1=2000, n2=2000, n_nodes=100, mask_frac=0.2) for i in range(32)]
    for p in cand:
        p["avail2"] = None
        p["desc2"], p["fv2"], p["angle2"] = cand[0]["desc2"], cand[0]["fv2"], cand[0]["angle2"]   # one frame, many KFs
    sb = {}
    for name, batch in (("pair", cand[:1]), ("batch32", cand)):
        msorb.search_by_bow(batch)
        kms = [msorb.search_by_bow(batch)[1] for _ in range(10)]
        t0 = time.perf_counter()
        for _ in range(10):
            out, _ = msorb.search_by_bow(batch)
        wall = (time.perf_counter() - t0) / 10
        t0 = time.perf_counter()
        for p, o in zip(batch, out):
            nm, m12, _ = orb_oracle.search_by_bow(p["desc1"], p["desc2"], p["valid1"], None, p["fv1"], p["fv2"], p["angle1"],
                                                  p["angle2"], 50, True, 0.7, True)
            assert nm == o[0] and m12.tolist() == o[1].tolist()
        dt = (time.perf_counter() - t0) / len(batch)
        sb[name] = dict(pairs=len(batch), kernel_ms=round(float(np.median(kms)), 4), wall_ms_python=round(wall * 1e3, 3),
                        matches_first=int(out[0][0]), cpu_oracle_ms_per_pair=round(dt * 1e3, 3))
    # the same with the candidate KeyFrames resident on the device (msorb_kf_store): per call only flags, work items and the frame move
    st = msorb.KeyFrameStore()
    kps0 = np.zeros(2000, msorb.KP_DTYPE)
    sc8 = np.array([1.2 ** i for i in range(8)], np.float32)
    ids = []
    for p in cand:
        k = kps0.copy()
        k["angle"] = p["angle1"]
        ids.append(st.add(k, p["desc1"], p["fv1"], sc8, sc8 * sc8))
    frame = dict(desc=cand[0]["desc2"], fv=cand[0]["fv2"], angle=cand[0]["angle2"])
    for name, idx in (("pair_resident", [0]), ("batch32_resident", list(range(32)))):
        pr = [dict(kf1=ids[i], kf2=-1, valid1=cand[i]["valid1"]) for i in idx]
        st.search_by_bow(pr, frame)
        kms = [st.search_by_bow(pr, frame)[1] for _ in range(10)]
        t0 = time.perf_counter()
        for _ in range(10):
            out, _ = st.search_by_bow(pr, frame)
        wall = (time.perf_counter() - t0) / 10
        ref, _ = msorb.search_by_bow([cand[i] for i in idx])
        assert all(o[0] == r[0] and o[1].tolist() == r[1].tolist() for o, r in zip(out, ref))
        sb[name] = dict(pairs=len(idx), kernel_ms=round(float(np.median(kms)), 4), wall_ms_python=round(wall * 1e3, 3),
                        wall_over_kernel=round(wall * 1e3 / float(np.median(kms)), 2))
    res["search_by_bow"] = sb
    # SearchForTriangulation: one CreateNewMapPoints pass = the new KeyFrame against 16 neighbours, 2000 features a side
    tri = [bmc.make_triangulation_pair(60 + i, n1=2000, n2=2000, n_nodes=100, mask_frac=0.35) for i in range(16)]
    st = {}
    for name, batch in (("pair", tri[:1]), ("neighbours16", tri)):
        msorb.search_for_triangulation(batch)
        kms = [msorb.search_for_triangulation(batch)[1] for _ in range(10)]
        t0 = time.perf_counter()
        for _ in range(10):
            out, _ = msorb.search_for_triangulation(batch)
        wall = (time.perf_counter() - t0) / 10
        t0 = time.perf_counter()
        for p, o in zip(batch, out):
            nm, m12 = orb_oracle.search_for_triangulation(p)
            assert nm == o[0] and m12.tolist() == o[1].tolist()
        dt = (time.perf_counter() - t0) / len(batch)
        st[name] = dict(pairs=len(batch), kernel_ms=round(float(np.median(kms)), 4), wall_ms_python=round(wall * 1e3, 3),
                        matches_first=int(out[0][0]), cpu_oracle_ms_per_pair=round(dt * 1e3, 3))
    for p in tri:
        for k in ("desc1", "fv1", "kp1"):
            p[k] = tri[0][k]
    cur = st_store_add = None
    store = msorb.KeyFrameStore()
    cur = store.add(tri[0]["kp1"], tri[0]["desc1"], tri[0]["fv1"], sc8, sc8 * sc8)
    nb = [store.add(p["kp2"], p["desc2"], p["fv2"], p["scale_factors2"], p["level_sigma2_2"]) for p in tri]
    for name, idx in (("pair_resident", [0]), ("neighbours16_resident", list(range(16)))):
        pr = [dict(kf1=cur, kf2=nb[i], valid1=tri[i]["valid1"], avail2=tri[i]["avail2"], stereo1=tri[i]["stereo1"],
                   stereo2=tri[i]["stereo2"], F12=tri[i]["F12"], ep=tri[i]["ep"]) for i in idx]
        store.search_for_triangulation(pr)
        kms = [store.search_for_triangulation(pr)[1] for _ in range(10)]
        t0 = time.perf_counter()
        for _ in range(10):
            out, _ = store.search_for_triangulation(pr)
        wall = (time.perf_counter() - t0) / 10
        ref, _ = msorb.search_for_triangulation([tri[i] for i in idx])
        assert all(o[0] == r[0] and o[1].tolist() == r[1].tolist() for o, r in zip(out, ref))
        st[name] = dict(pairs=len(idx), kernel_ms=round(float(np.median(kms)), 4), wall_ms_python=round(wall * 1e3, 3),
                        wall_over_kernel=round(wall * 1e3 / float(np.median(kms)), 2))
    store.close()
    res["search_for_triangulation"] = st
    print(json.dumps(res))


if __name__ == "__main__":
    main()
