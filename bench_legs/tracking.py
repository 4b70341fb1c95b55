"""configs[2]: the front-end of a tracking frame — fused device chains, the reference-KeyFrame leg and their CPU oracle legs."""
import os
import sys
import time

import numpy as np

from . import ROOT, KITTI_MB, KITTI_MBF, self_check, oracle_module, tests_dir


def tracking_loop_leg(msorb, synth, torch, ex, cfg, base, counts_h, d_kps, d_desc, d_ur, dev, local, args, m_points=4096):
    """BASELINE configs[2] ("extract + ORBmatcher::SearchByProjection inside the full Tracking loop"): (a) one frame at a time
    through the C ABI from HOST images — msorb_track_frontend (one call, one synchronisation) and msorb_extract_stereo_frame +
    msorb_search_local_points (two calls: the pose estimate of TrackWithMotionModel sits between them in the reference) —,
    (b) the device part for a batch of frames (msorb_track_batch on the extraction outputs already in HBM), with the windowed
    Hamming rate = distances evaluated by the window search / its kernel time."""
    cap = d_kps.shape[1]
    scale = ex.GetScaleFactors()
    n_frames = min(len(counts_h) // 2, 64)
    kps_h = d_kps[0:2 * n_frames:2].cpu().numpy()
    desc_h = d_desc[0:2 * n_frames:2].cpu().numpy()
    d_dp = None
    d_ur2, d_dp, _, _ = msorb.stereo_matches_batch(ex, counts_h, d_kps, d_desc, KITTI_MB, KITTI_MBF)
    dp_h = d_dp[:n_frames].cpu().numpy()
    cam = synth.KITTI_CAM
    bounds = (0.0, float(cfg["cols"]), 0.0, float(cfg["rows"]))
    maps, frusta = [], []
    for b in range(n_frames):
        n = int(counts_h[2 * b])
        k = kps_h[b, :n].copy().view(msorb.KP_DTYPE).reshape(-1)
        # spars_frac = 0: MapPoint::mbSparsified is initialised false and never set in MS-SLAM (only KeyFrame::mbSparsified is,
        # KeyFrame.cc:359), so the bypass of ORBmatcher.cc:88 — a point that overwrites an occupied keypoint — never fires in the
        # reference; the parity tests keep exercising it (5 % of the points), the timed workload does not
        mp = synth.local_map(9000 + b, k, desc_h[b, :n], dp_h[b, :n], scale, m_points, spars_frac=0.0)
        maps.append(mp)
        frusta.append(msorb.Frustum.make(mp["Rcw"], mp["tcw"], mp["Ow"], cam["fx"], cam["fy"], cam["cx"], cam["cy"], bounds, cam["mbf"],
                                         float(np.log(np.float32(cfg["scale"]))), cfg["nlevels"]))
    d_mp = {k: torch.from_numpy(np.stack([mp[k] for mp in maps])).to(dev).contiguous()
            for k in ("pos_w", "normal", "max_distance", "min_distance", "flags", "desc")}
    th = 1.0   # Tracking::SearchLocalPoints: th = 1 in the steady state (Tracking.cc:3363-3386)
    r = msorb.track_batch(d_kps, d_desc, d_ur2, counts_h, 2, bounds, scale, frusta, d_mp, th, count_pairs=True, device=local)
    n_eval = r["n_pairs"]
    ms = np.array([msorb.track_batch(d_kps, d_desc, d_ur2, counts_h, 2, bounds, scale, frusta, d_mp, th, device=local)["ms"] for _ in range(15)])
    ms_grid, ms_frustum, ms_window = [float(x) for x in np.median(ms, axis=0)]
    g = n_eval / (ms_window * 1e-3) / 1e9
    ceil_valu = 1024 * 2.4e9 * 64 / (8 * 7.91) / 1e9   # 8 x (v_xor + v_bcnt) in a mixed stream, tools/valu_ubench2.hip
    in_view = int(r["in_view"].sum().item())
    with_cand = int((r["topk_idx"][:, :, 0] >= 0).sum().item())
    # (a) per frame, host images in, host features + matches out
    left, right = base[0], base[1]
    run = msorb.TrackFrontendRunner(ex, left, right, KITTI_MB, KITTI_MBF, frusta[0], maps[0], th, device=local)
    for _ in range(5):
        run.one_call()
        run.two_calls()
    t1, t2 = [], []
    for _ in range(60):
        t0 = time.perf_counter(); nm1 = run.one_call(); t1.append(time.perf_counter() - t0)
    for _ in range(60):
        t0 = time.perf_counter(); nm2 = run.two_calls(); t2.append(time.perf_counter() - t0)
    # TrackWithMotionModel's half of the frame (a14): the last frame's points around this frame's keypoints, th = 7 (stereo)
    n0 = int(counts_h[0])
    k0 = kps_h[0, :n0].copy().view(msorb.KP_DTYPE).reshape(-1)
    last, q_cw, t_cw, fwd, bwd = synth.last_frame(9500, k0, desc_h[0, :n0], dp_h[0, :n0])
    mm = msorb.MotionModel.make(q_cw, t_cw, cam["fx"], cam["fy"], cam["cx"], cam["cy"], cam["mbf"], fwd, bwd)
    th_mm = 7.0   # Tracking.cc:2847-2850
    mrun = msorb.MotionFrontendRunner(ex, left, right, KITTI_MB, KITTI_MBF, mm, last, last["obs"], th_mm, device=local)
    mrun.attach_local_points(frusta[0], maps[0], th)
    for _ in range(5):
        mrun.one_call(); mrun.separate_calls(); mrun.frame_total()
    tm1, tm3, tms, tft = [], [], [], []
    for _ in range(60):
        t0 = time.perf_counter(); nmm1 = mrun.one_call(); tm1.append(time.perf_counter() - t0)
    mm_cur = mrun.cur_mp[:n0].copy()
    for _ in range(60):
        t0 = time.perf_counter(); nmm3 = mrun.separate_calls(); tm3.append(time.perf_counter() - t0)
    self_check(nmm1 == nmm3 and np.array_equal(mm_cur, mrun.cur_mp[:n0]),
               "tracking_loop: msorb_track_frontend_motion and the separate calls return different matches")
    for _ in range(60):
        t0 = time.perf_counter(); mrun.search_only(); tms.append(time.perf_counter() - t0)
    for _ in range(60):
        t0 = time.perf_counter(); nm_a, nm_b = mrun.frame_total(); tft.append(time.perf_counter() - t0)
    self_check(nm_a == nmm1, "tracking_loop: the motion-model search of frame_total differs")
    n_kp = int(run.nl.value + run.nr.value)
    self_check(nm1 == nm2, "tracking_loop: msorb_track_frontend and the two-call form return different match counts")
    out = {"what": "configs[2]: front-end of one tracking frame (Frame.cc:119-137 + Tracking::SearchLocalPoints, Tracking.cc:3343-3388) as a "
                   "device-resident chain; local map of %d points per frame (70 %% on keypoint rays, descriptors <= 40 bits off), th = 1" % m_points,
           "per_frame": {"ms_one_call": round(float(np.median(t1)) * 1e3, 4), "ms_two_calls": round(float(np.median(t2)) * 1e3, 4),
                         "keypoints": n_kp, "matches": int(nm1), "same_matches_both_ways": bool(nm1 == nm2),
                         "window_rounds": int(run.rounds.value),
                         "note": "wall time through the C ABI (ctypes call included), host images in, host features + matches out; "
                                 "one_call = msorb_track_frontend, two_calls = msorb_extract_stereo_frame + msorb_search_local_points"},
           "motion_model": {"what": "TrackWithMotionModel's search (Tracking.cc:2833-2870 -> ORBmatcher::SearchByProjection(Current, Last, th, "
                                    "bMono), ORBmatcher.cc:1941-2152) with the projection on the device: last-frame table of %d keypoints, "
                                    "%d of them with a map point, th = %g" % (n0, int(last["has_point"].sum()), th_mm),
                            "ms_frame_and_search_one_call": round(float(np.median(tm1)) * 1e3, 4),
                            "ms_frame_and_search_separate_calls": round(float(np.median(tm3)) * 1e3, 4),
                            "ms_search_only": round(float(np.median(tms)) * 1e3, 4), "matches": int(nmm1),
                            "same_matches_both_ways": True,
                            "note": "one_call = msorb_frame_set_last_points + msorb_track_frontend_motion (host images and the host "
                                    "table in, features + cur_mp out); search_only = msorb_search_last_frame on the resident table "
                                    "(the retry at 2 * th of Tracking.cc:2861-2868 costs this)"},
           "per_frame_total": {"ms": round(float(np.median(tft)) * 1e3, 4), "motion_model_matches": int(nm_a), "local_map_matches": int(nm_b),
                               "what": "both device calls of ONE tracking frame in the order Tracking::Track runs them: (1) Frame::Frame + "
                                       "TrackWithMotionModel's SearchByProjection (msorb_frame_set_last_points + msorb_track_frontend_motion), "
                                       "(2) TrackLocalMap's SearchLocalPoints (msorb_search_local_points: isInFrustum + SearchByProjection over "
                                       "%d local map points; the motion-model matches stay in the frame as in Tracking::Track: their "
                                       "keypoints are occupied and their points — rows of the local map — are already seen and skipped).  The host's PoseOptimization between and after them is NOT included (g2o, out "
                                       "of scope); wall time through the C ABI from host images" % m_points},
           "batched": {"frames": n_frames, "map_points_per_frame": m_points, "ms_grid": round(ms_grid, 4), "ms_frustum_queries": round(ms_frustum, 4),
                       "ms_window_search": round(ms_window, 4), "ms_per_frame": round((ms_grid + ms_frustum + ms_window) / n_frames, 5),
                       "points_in_view": in_view, "points_with_candidates": with_cand,
                       "note": "device part only (frame_grid_kernel, local_points_kernel, window_topk_kernel) on features already in HBM"},
           "windowed_hamming": {"pairs_evaluated": int(n_eval), "gpairs_per_s": round(g, 3), "kernel": "window_topk_kernel",
                                "valu_popcount_ceiling_gpairs_per_s": round(ceil_valu, 1), "frac": round(g / ceil_valu, 5),
                                "bound": "grid walk + dependent gathers (cell -> index -> keypoint -> descriptor), a handful of "
                                         "distances per query: latency, not VALU issue or HBM"},
           "_cpu": (maps[0], frusta[0], kps_h[0, :int(counts_h[0])].copy().view(msorb.KP_DTYPE).reshape(-1), desc_h[0, :int(counts_h[0])],
                    d_ur2[0, :int(counts_h[0])].cpu().numpy(), bounds, scale, th,
                    int((r["topk_idx"][0, :, 0] >= 0).sum().item()))}
    # the same frame through the UNCHANGED call sites (drop-in classes, host-projected a14): bench_legs/unchanged.py
    from . import optional_leg
    from .unchanged import unchanged_callers_leg
    out["unchanged_callers"] = optional_leg("tracking_loop.unchanged_callers", unchanged_callers_leg, msorb, synth, cfg, left, right, k0,
                                            desc_h[0, :n0].copy(), d_ur2[0, :n0].cpu().numpy(), dp_h[0, :n0].copy(), np.asarray(scale, np.float32),
                                            maps[0], frusta[0], last, th_mm, th, args.cpu_pairs > 0, device=local)
    out["_cpu_mm"] = (last, q_cw, t_cw, bool(fwd), bool(bwd), th_mm, int(nmm1), mm_cur, k0, desc_h[0, :n0].copy(),
                      d_ur2[0, :n0].cpu().numpy(), bounds, scale)
    run.close()
    mrun.close()
    return out


def reference_keyframe_leg(msorb, cpu):
    """TrackReferenceKeyFrame's device part per frame (Tracking.cc:2703-2713): Frame::ComputeBoW (Frame.cc:670-677: DBoW2
    transform of the frame's descriptors, levelsup 4) + ORBmatcher(0.7).SearchByBoW(pReferenceKF, Frame) against a KeyFrame
    resident on the device.  ORBvoc-shaped synthetic vocabulary (k = 10, L = 6: the real ORBvoc.txt is a missing blob of the
    reference); 2000 descriptors = noisy vocabulary leaves, the KeyFrame's = the frame's with up to 25 bits flipped."""
    tests_dir()
    import bow_cases
    voc = bow_cases.make_vocabulary(0, k=10, L=6, stop_frac=0.01)
    dev = msorb.Vocabulary(voc["k"], voc["L"], 0, 0, voc["parent"], voc["is_leaf"], voc["descriptors"], voc["weights"])
    rng = np.random.default_rng(3)
    n = 2000
    d_frame = bow_cases.make_features(1, voc, n)
    d_kf = bow_cases._flip_bits(rng, d_frame, rng.integers(0, 26, n))
    kps = np.zeros(n, msorb.KP_DTYPE)
    kps["angle"] = rng.uniform(0, 360, n)
    a_kf = ((kps["angle"] + rng.normal(0, 5, n)) % 360).astype(np.float32)
    kkf = kps.copy()
    kkf["angle"] = a_kf
    sc8 = np.array([1.2 ** i for i in range(8)], np.float32)
    fv = lambda r: (r["fv_node"], r["fv_begin"], r["fv_feat"])
    store = msorb.KeyFrameStore()
    kid = store.add(kkf, d_kf, fv(dev.transform(d_kf)), sc8, sc8 * sc8)
    valid1 = np.ones(n, np.uint8)
    for _ in range(5):
        rb = dev.transform(d_frame)
        store.search_by_bow([dict(kf1=kid, kf2=-1, valid1=valid1)], dict(desc=d_frame, fv=fv(rb), angle=kps["angle"]))
    tb, ts = [], []
    for _ in range(40):
        t0 = time.perf_counter()
        rb = dev.transform(d_frame)
        t1 = time.perf_counter()
        out, _ = store.search_by_bow([dict(kf1=kid, kf2=-1, valid1=valid1)], dict(desc=d_frame, fv=fv(rb), angle=kps["angle"]))
        tb.append(t1 - t0); ts.append(time.perf_counter() - t1)
    res = {"what": "TrackReferenceKeyFrame's device part per frame (Tracking.cc:2703-2713): Frame::ComputeBoW (msorb_bow_transform, host "
                   "arrays in and out) + SearchByBoW(pReferenceKF, Frame) against a resident KeyFrame (msorb_search_by_bow_kf); "
                   "synthetic ORBvoc-shaped vocabulary (k 10, L 6), 2000 descriptors a side",
           "ms_compute_bow": round(float(np.median(tb)) * 1e3, 4), "ms_search_by_bow": round(float(np.median(ts)) * 1e3, 4),
           "words": int(len(rb["bow_word"])), "nodes": int(len(rb["fv_node"])), "matches": int(out[0][0])}
    if cpu:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import orb_oracle
        orc = orb_oracle.OracleVocabulary(voc["k"], voc["L"], 0, 0, voc["parent"], voc["is_leaf"], voc["descriptors"], voc["weights"])
        rk = orc.transform(d_kf)
        t0 = time.perf_counter()
        for _ in range(5):
            rf = orc.transform(d_frame)
        t1 = time.perf_counter()
        for _ in range(5):
            nm, m12, _ = orb_oracle.search_by_bow(d_kf, d_frame, valid1, None, fv(rk), fv(rf), a_kf, kps["angle"], 50, True, 0.7, True)
        t2 = time.perf_counter()
        same = (rf["bow_word"].tolist() == rb["bow_word"].tolist() and rf["bow_value"].tobytes() == rb["bow_value"].tobytes() and
                nm == out[0][0] and m12.tolist() == out[0][1].tolist())
        self_check(same, "tracking_loop.reference_keyframe: ComputeBoW / SearchByBoW differ from the CPU oracle")
        res["cpu_baseline"] = {"ms_compute_bow": round((t1 - t0) / 5 * 1e3, 4), "ms_search_by_bow": round((t2 - t1) / 5 * 1e3, 4), "cores": 1,
                               "kind": "port", "gpu_matches_cpu": True}
    store.close()
    dev.close()
    return res


def tracking_cpu_leg(tracking, msorb):
    """CPU oracle leg of configs[2]'s matcher half: isInFrustum + SearchByProjection over one frame's local map, 1 thread (the
    reference's tracking thread), and the cross-check of the device chain's matches against it."""
    orb_oracle = oracle_module()
    mp, fr, kps, desc, ur, bounds, scale, th, _ = tracking["_cpu"]
    rf = orb_oracle.OracleFrame(kps, desc, ur, bounds, scale)
    t0 = time.perf_counter()
    reps = 20
    for _ in range(reps):
        r = orb_oracle.is_in_frustum(fr, mp["pos_w"], mp["normal"], mp["max_distance"], mp["min_distance"], 0.5)
        tab = dict(track_in_view=(r["track_in_view"].astype(bool) & mp["visit"].astype(bool)).astype(np.uint8), bad=mp["bad"],
                   sparsified=mp["sparsified"], proj_x=r["proj_x"], proj_y=r["proj_y"], proj_xr=r["proj_xr"], track_depth=r["track_depth"],
                   level=r["level"], view_cos=r["view_cos"], desc=mp["desc"], obs=mp["obs"])
        frame_mp = np.full(len(kps), -1, np.int32)
        nm = rf.SearchByProjection_mps(tab, frame_mp, th)
    dt = (time.perf_counter() - t0) / reps
    f = msorb.Frame(kps, desc, ur, bounds, scale)
    g_mp = np.full(len(kps), -1, np.int32)
    g_nm, _ = msorb.search_local_points(f, fr, mp, g_mp, th)
    f.close()
    self_check(g_nm == nm and np.array_equal(g_mp, frame_mp), "tracking_loop: msorb_search_local_points differs from the CPU oracle")
    # motion-model half on the CPU: projection (orc_project_last_frame) + SearchByProjection(Current, Last), 1 thread
    last, q_cw, t_cw, fwd, bwd, th_mm, g_nmm, g_cur, k0, d0, ur0, bounds_mm, scale_mm = tracking.pop("_cpu_mm")
    omm = orb_oracle.MotionModel()
    omm.q[:] = [float(v) for v in q_cw]
    omm.t[:] = [float(v) for v in t_cw]
    from msorb import synth as _synth
    cam = _synth.KITTI_CAM
    omm.fx, omm.fy, omm.cx, omm.cy, omm.mbf = cam["fx"], cam["fy"], cam["cx"], cam["cy"], cam["mbf"]
    rf2 = orb_oracle.OracleFrame(k0, d0, ur0, bounds_mm, scale_mm)
    t0 = time.perf_counter()
    for _ in range(reps):
        valid, u, v, urp = orb_oracle.project_last_frame(omm, bounds_mm, last["has_point"], last["pos_w"])
        tab = dict(valid=valid, u=u, v=v, ur=urp, octave=last["octave"], angle=last["angle"], desc=last["desc"],
                   mp=np.arange(len(valid), dtype=np.int32), obs=last["obs"])
        c_cur = np.full(len(k0), -1, np.int32)
        c_nmm = rf2.SearchByProjection_frames(tab, c_cur, th_mm, fwd, bwd, True)
    dtm = (time.perf_counter() - t0) / reps
    self_check(c_nmm == g_nmm and np.array_equal(c_cur, g_cur), "tracking_loop: msorb_track_frontend_motion differs from the CPU oracle")
    tracking["motion_model"]["cpu_baseline"] = {"ms_per_frame": round(dtm * 1e3, 4), "cores": 1, "kind": "port", "matches": int(c_nmm),
                                                "sample": f"projection + SearchByProjection(Current, Last) over {len(valid)} last-frame "
                                                          f"keypoints, oracle, {reps} repetitions", "gpu_matches_cpu": True}
    tracking["cpu_baseline"] = {"ms_per_frame_matcher_half": round(dt * 1e3, 4), "cores": 1, "kind": "port", "matches": int(nm),
                                "sample": f"isInFrustum + SearchByProjection over {len(mp['obs'])} map points x {len(kps)} keypoints, oracle, "
                                          f"{reps} repetitions", "gpu_matches_cpu": bool(g_nm == nm and np.array_equal(g_mp, frame_mp))}
