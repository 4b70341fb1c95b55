#!/usr/bin/env python3
"""bench.py — ORB front-end throughput on MI355X (BASELINE.json metric, configs[1] workload).

A "step" = one pass of the hot path (ORBextractor::operator(): pyramid -> FAST+NMS -> quadtree -> blur ->
IC-angle -> rBRIEF) over one batch of synthetic KITTI-00-like stereo pairs (1241x376, 2000 features/frame),
inputs already resident in HBM.  value = keypoints returned per second, whole job.

  python bench.py --gpus N --steps K --warmup W [--pairs B]

N=1: both eyes of every pair on the one GPU.  N>1 (launched by torch.distributed.run, one rank per GPU):
stereo left/right split — even ranks extract the left eyes, odd ranks the right eyes of their pair group,
then the odd rank sends counts/keypoints/descriptors to its even partner over RCCL (xGMI point-to-point);
per-GPU work is fixed (weak scaling).

Prints ONE JSON line on rank 0 with `roofline` (dominant kernel, HIP-event timed) and `cpu_baseline`
(the CPU oracle timed on a bounded sample of the same workload on this box's host cores).
"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
KITTI_MBF = 386.1448            # Camera.bf of Examples/Stereo/KITTI00-02.yaml
KITTI_MB = KITTI_MBF / 718.856  # mb = mbf / fx
sys.path[:0] = [os.path.join(ROOT, "ms-slam_amd")]

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md)


def self_check(ok, what):
    """Every cross-check the line reports (`*_matches_cpu`, `identical_results`, `same_matches_*`) is enforced: a bench line is
    only printed when all of them hold, so a `false` can never appear as a result."""
    if not ok:
        raise AssertionError("bench self-check failed: " + what)


def live_pmc(kernel_substr, counters=("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU"), timeout_s=150):
    """roofline.traffic measured in THIS run: one rocprofv3 --pmc pass per counter (separate passes, kernel trace only — the
    guide's recipe) over a child `bench.py --steps 2 --warmup 1 --lean --isolated --no-pmc` (the same batch, every kernel alone
    on the GPU), per-launch average of the dominant kernel.  -> {counter: value} or None when rocprofv3 is missing, refuses the
    counter or does not finish (the line then falls back to the committed PMC summary and says so)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    rp = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rp):
        return None
    out = {}
    for ctr in counters:
        d = tempfile.mkdtemp(prefix="msorb_pmc_", dir="/tmp")
        try:
            env = dict(os.environ, TMPDIR="/tmp")
            subprocess.run([rp, "--pmc", ctr, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "p", "--", sys.executable,
                            os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--lean", "--isolated", "--no-pmc", "--cpu-pairs", "0"],
                           cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=timeout_s, check=False)
            files = glob.glob(os.path.join(d, "**", "p_counter_collection.csv"), recursive=True)
            if not files:
                return None
            vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(files[0]))
                    if r["Counter_Name"] == ctr and kernel_substr in r["Kernel_Name"]]
            if not vals:
                return None
            out[ctr] = sum(vals) / len(vals)
        except Exception:
            return None
        finally:
            shutil.rmtree(d, ignore_errors=True)
    return out


def level_bytes(cfg):
    """Algorithmic bytes per image: sum of level pixels etc."""
    import math
    sc = np.float32(1.0)
    px = []
    for l in range(cfg["nlevels"]):
        inv = np.float32(1.0) / sc
        w = int(np.rint(np.float32(cfg["cols"]) * inv))
        h = int(np.rint(np.float32(cfg["rows"]) * inv))
        px.append(w * h)
        sc = np.float32(np.float64(sc) * np.float64(np.float32(cfg["scale"])))
    return px


def cpu_baseline(cfg, sample_pairs, seed0):
    """Oracle (CPU restatement, kind=port) on `sample_pairs` stereo pairs, 2 threads = one per eye like
    the reference (Frame.cc:122-125)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import orb_oracle
    from msorb import synth
    exs = [orb_oracle.OracleExtractor(cfg["nfeatures"], cfg["scale"], cfg["nlevels"], cfg["ini_th"], cfg["min_th"])
           for _ in range(2)]
    uniq = [synth.stereo_pair(seed0 + i, cfg["rows"], cfg["cols"]) for i in range(min(sample_pairs, 16))]
    pairs = [uniq[i % len(uniq)] for i in range(sample_pairs)]   # the oracle recomputes every image: repeats cost the same
    counts = [0, 0]

    def eye(e):
        for p in pairs:
            _, kps, _ = exs[e](p[e])
            counts[e] += len(kps)

    t0 = time.perf_counter()
    th = [threading.Thread(target=eye, args=(e,)) for e in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    dt = time.perf_counter() - t0
    out = dict(value=round(sum(counts) / dt / 1e6, 5), unit="Mkeypoints/s", cores=2, kind="port",
               sample=f"{sample_pairs} KITTI-like stereo pairs, oracle/ (scalar C++ restatement, not OpenCV SIMD), "
                      f"2 threads (one per eye), {dt:.1f} s",
               host_cpus=os.cpu_count())
    # the same port scaled over the host's cores by frame-level parallelism: one extractor object per
    # thread, 2 images each — what an offline CPU pipeline could reach on this box
    nthr = max(2, min(os.cpu_count() or 2, 128))
    exs2 = [orb_oracle.OracleExtractor(cfg["nfeatures"], cfg["scale"], cfg["nlevels"], cfg["ini_th"], cfg["min_th"])
            for _ in range(nthr)]
    imgs = [pairs[i % len(pairs)][i % 2] for i in range(4)]
    tot = [0] * nthr

    def worker(i):
        for k in range(2):
            _, kps, _ = exs2[i](imgs[(i + k) % 4])
            tot[i] += len(kps)

    t0 = time.perf_counter()
    th = [threading.Thread(target=worker, args=(i,)) for i in range(nthr)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    dt2 = time.perf_counter() - t0
    out["all_cores"] = dict(value=round(sum(tot) / dt2 / 1e6, 4), unit="Mkeypoints/s", cores=nthr,
                            sample=f"{2 * nthr} images over {nthr} threads, {dt2:.1f} s")
    # BASELINE.json configs[0]: "EuRoC MH_01 stereo, CPU ORBextractor at 1000 features/frame (reference path, no GPU)" — the
    # same port on EuRoC-like 752x480 pairs, 2 threads (one per eye)
    ec = synth.EUROC
    exs3 = [orb_oracle.OracleExtractor(ec["nfeatures"], ec["scale"], ec["nlevels"], ec["ini_th"], ec["min_th"]) for _ in range(2)]
    ne = max(2, min(12, sample_pairs // 8))
    epairs = [synth.stereo_pair(seed0 + 500 + i, ec["rows"], ec["cols"]) for i in range(min(ne, 4))]
    ecount = [0, 0]

    def eeye(e):
        for i in range(ne):
            _, kps, _ = exs3[e](epairs[i % len(epairs)][e])
            ecount[e] += len(kps)

    t0 = time.perf_counter()
    th = [threading.Thread(target=eeye, args=(e,)) for e in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    dt3 = time.perf_counter() - t0
    out["configs0_euroc_1000"] = dict(value=round(sum(ecount) / dt3 / 1e6, 5), unit="Mkeypoints/s", cores=2, kind="port",
                                      ms_per_stereo_frame=round(dt3 / ne * 1e3, 2),
                                      sample=f"{ne} EuRoC-like 752x480 stereo pairs at 1000 features, 2 threads, {dt3:.1f} s")
    return out


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
        mp = synth.local_map(9000 + b, k, desc_h[b, :n], dp_h[b, :n], scale, m_points)
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
                                       "%d local map points).  The host's PoseOptimization between and after them is NOT included (g2o, out "
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
    sys.path.insert(0, os.path.join(ROOT, "tests"))
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


def tracking_cpu_leg(tracking, msorb, oracle_dir):
    """CPU oracle leg of configs[2]'s matcher half: isInFrustum + SearchByProjection over one frame's local map, 1 thread (the
    reference's tracking thread), and the cross-check of the device chain's matches against it."""
    sys.path.insert(0, oracle_dir)
    import orb_oracle
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


def split_self_validation(msorb, torch, dist, stereo_split, rank, world, eye, half, ex_own, ex_other, make_ex, images, other_images,
                          mine, theirs, rank_value, dev):
    """Untimed, after the timed region of a --gpus N run: (1) who took part (ranks, devices, backend), (2) one more exchange, after
    which every rank checks the features it RECEIVED for its pairs against a local extraction of the same images (it holds the
    other eye's images of the pairs it joins) — bit for bit —, and that the stereo association on (own, gathered) features
    equals the association on (own, locally extracted) ones: the split path gives the single-GPU result; (3) per-rank rates.
    An assertion failure here aborts the run: a wrong 2-GPU number is never printed."""
    backend = dist.get_backend()
    info = [None] * world
    dist.all_gather_object(info, {"rank": rank, "device": torch.cuda.get_device_name(dev), "cuda_index": dev.index,
                                  "mkeypoints_per_s": round(rank_value / 1e6, 3)})
    ones = torch.ones(1, dtype=torch.int32, device=dev)
    dist.all_reduce(ones)
    # fresh extraction of this rank's images + exchange
    counts, _, _, _ = ex_own.extract_batch(images, (0, 0), out=(mine.kps, mine.desc))
    mine.counts.copy_(torch.from_numpy(counts))
    works = stereo_split.swap_halves_async(dist, rank, world, mine, theirs)
    stereo_split.finish(works)
    same_features = same_assoc = None
    if works:
        ex_chk = make_ex()
        try:
            c_loc, _, k_loc, d_loc = ex_chk.extract_batch(other_images)          # the other eye of my pairs, extracted HERE
            got_c = theirs.counts.cpu().numpy()
            same_features = bool(np.array_equal(got_c, c_loc[:half]))
            for i in range(half):
                n = int(c_loc[i])
                same_features = same_features and bool(torch.equal(theirs.kps[i, :n], k_loc[i, :n]) and torch.equal(theirs.desc[i, :n], d_loc[i, :n]))
            own = (mine.counts[:half], mine.kps[:half], mine.desc[:half])
            got = (theirs.counts, theirs.kps, theirs.desc)
            loc = (torch.from_numpy(np.ascontiguousarray(c_loc[:half])).to(dev), k_loc[:half], d_loc[:half])
            ex_other.pyramid_batch(other_images)
            if eye == 0:
                ur_split = msorb.stereo_matches_split(ex_own, ex_other, *own, *got, KITTI_MB, KITTI_MBF)[0]
                ur_local = msorb.stereo_matches_split(ex_own, ex_chk, *own, *loc, KITTI_MB, KITTI_MBF)[0]
            else:
                ur_split = msorb.stereo_matches_split(ex_other, ex_own, *got, *own, KITTI_MB, KITTI_MBF)[0]
                ur_local = msorb.stereo_matches_split(ex_chk, ex_own, *loc, *own, KITTI_MB, KITTI_MBF)[0]
            same_assoc = bool(torch.equal(ur_split, ur_local))
            matched = int((ur_split > 0).sum().item())
        finally:
            ex_chk.close()
        assert same_features, f"rank {rank}: the gathered features differ from a local extraction of the same images"
        assert same_assoc, f"rank {rank}: the split stereo association differs from the single-GPU association"
    flags = [None] * world
    dist.all_gather_object(flags, {"rank": rank, "gathered_features_equal_local": same_features, "split_association_equals_local": same_assoc,
                                   "matched": matched if works else None})
    return {"backend": backend, "rccl": backend == "nccl", "ranks_seen": int(ones.item()), "ranks": info,
            "bytes_exchanged_per_step_and_rank": {"sent": sum(t[half:].numel() * t.element_size() for t in mine.tensors()),
                                                  "received": theirs.nbytes()},
            "checks": flags}


def host_fed_leg(msorb, torch, exs, host_images, dev, cfg, pitch, steps=12):
    """The same extraction fed from pinned HOST memory: the upload of batch k+1 (one hipMemcpy2D-shaped copy into the 64-byte
    pitch planes, on a copy stream) runs while batch k is extracted.  What a pipeline that receives frames in host memory gets,
    next to `value` (inputs already in HBM)."""
    n = host_images.shape[0]
    pinned = torch.from_numpy(np.ascontiguousarray(host_images)).pin_memory()
    bufs = [torch.zeros((n, cfg["rows"], pitch), dtype=torch.uint8, device=dev) for _ in range(2)]
    views = [b[:, :, :cfg["cols"]] for b in bufs]
    cs = torch.cuda.Stream(device=dev)
    n_up = int(os.environ.get("MSORB_BENCH_UPLOAD_STREAMS", "2"))   # the batch as two slices on two copy streams (two DMA engines:
    # 51.7 instead of 46.4 GB/s; four streams: no more)
    css = [cs] + [torch.cuda.Stream(device=dev) for _ in range(n_up - 1)]
    outs = [None, None]
    for e in exs:
        e.set_overlap(1, True)

    def upload(k):
        for i, c in enumerate(css):
            a, b = n * i // n_up, n * (i + 1) // n_up
            with torch.cuda.stream(c):
                views[k][a:b].copy_(pinned[a:b], non_blocking=True)
        for c in css[1:]:
            cs.wait_stream(c)

    upload(0)
    cs.synchronize()
    t_up = time.perf_counter()
    upload(1)
    cs.synchronize()
    t_up = time.perf_counter() - t_up
    kp, t0 = 0, None
    for k in range(steps + 2):
        b = k & 1
        exs[b].extract_batch_submit(views[b], (0, 0), out=outs[b])
        outs[b] = exs[b]._pending[2]
        if k >= 1:
            counts, _, _, _ = exs[b ^ 1].extract_batch_wait()      # batch k-1 done: its buffer is free ...
            if k >= 3:
                kp += int(counts.sum())
            upload(b ^ 1)                                          # ... for the upload of batch k+1, under batch k's kernels
        if k == 1:
            cs.synchronize()
            t0 = time.perf_counter()                               # steady state from here: every step = one upload + one batch
        else:
            cs.synchronize()
    counts, _, _, _ = exs[(steps + 1) & 1].extract_batch_wait()
    kp += int(counts.sum())
    dt = time.perf_counter() - t0
    nbytes = pinned.numel()
    return {"what": "extract+describe with every batch uploaded from pinned host memory (pitched copy, two slices on two copy streams) while the previous one is extracted",
            "mkeypoints_per_s": round(kp / dt / 1e6, 2), "ms_per_step": round(dt / (steps + 0) * 1e3, 4),
            "upload_ms_per_batch": round(t_up * 1e3, 4), "upload_gbs": round(nbytes / t_up / 1e9, 2), "bytes_per_batch": int(nbytes),
            "bound": "PCIe (the upload of a batch takes longer than its kernels)"}


def per_frame_leg(msorb, ex, left, right):
    """What an unchanged Frame.cc caller sees per frame (host cv::Mat in, host keypoints / descriptors out; the drop-in class adds
    the cv::Mat / std::vector conversions, ~0.01-0.04 ms, tools/latency_class.cc): msorb_extract on one image, and both eyes +
    ComputeStereoMatches in one call (msorb_extract_stereo), buffers prepared once."""
    import ctypes as C
    L = ex.L
    cap = ex.capacity
    rows, cols = left.shape
    left, right = np.ascontiguousarray(left), np.ascontiguousarray(right)
    kl, kr = np.zeros(cap, msorb.KP_DTYPE), np.zeros(cap, msorb.KP_DTYPE)
    dl, dr = np.zeros((cap, 32), np.uint8), np.zeros((cap, 32), np.uint8)
    ur, dp = np.zeros(cap, np.float32), np.zeros(cap, np.float32)
    n, mono, nl, nr, oob = C.c_int(), C.c_int(), C.c_int(), C.c_int(), C.c_int()
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    one = (ex.h, p(left), rows, cols, cols, 0, 0, p(kl), p(dl), cap, C.byref(n), C.byref(mono))
    L.msorb_extract_stereo.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_size_t, C.c_float, C.c_float,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                       C.c_void_p, C.c_void_p]
    st = (ex.h, p(left), p(right), rows, cols, cols, cols, KITTI_MB, KITTI_MBF, p(kl), p(dl), C.byref(nl), p(kr), p(dr), C.byref(nr), cap,
          p(ur), p(dp), C.byref(oob))
    t1, t2 = [], []
    for i in range(45):
        t0 = time.perf_counter(); L.msorb_extract(*one); t1.append(time.perf_counter() - t0)
    for i in range(45):
        t0 = time.perf_counter(); L.msorb_extract_stereo(*st); t2.append(time.perf_counter() - t0)
    m1, m2 = float(np.median(t1[5:])), float(np.median(t2[5:]))
    return {"what": "one frame at a time through the C ABI from host images (B = 1): what the drop-in ORBextractor::operator() costs",
            "ms_one_image": round(m1 * 1e3, 4), "ms_stereo_frame_one_call": round(m2 * 1e3, 4),
            "keypoints_stereo_frame": int(nl.value + nr.value), "mkeypoints_per_s_stereo_frame": round((nl.value + nr.value) / m2 / 1e6, 2)}


def sparsification_leg(msorb, cpu):
    """BASELINE configs[4]: the per-window constraint-matrix build of MapSparsification::Sparsifying (MapSparsification.cc:58-151)
    on a 4Seasons-like sliding window — 30 keyframes x 2000 slots, half of them tracked, 6000 map points, 100 keyframes outside
    the window —: msorb_visibility_csr through the C ABI (host arrays in, CSR out), every buffer prepared once."""
    import ctypes as C
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import sparsify_cases as sc
    w = sc.window(11)
    L = msorb.lib()
    arrs = {k: np.ascontiguousarray(w[k], np.uint8 if k == "kf_in_window" else np.int32) for k in
            ("kf_slot_begin", "slot_point", "slot_cell", "point_nobs", "obs_begin", "obs_kf", "kf_in_window", "kf_num_mps")}
    K, S, P, KT = len(arrs["kf_slot_begin"]) - 1, len(arrs["slot_point"]), len(arrs["point_nobs"]), len(arrs["kf_in_window"])
    cc, cr, cn = S + 1, S + K + KT + 1, 2 * S + len(arrs["obs_kf"]) + 1
    col_point, obj = np.zeros(cc, np.int32), np.zeros(cc, np.float32)
    row_begin, row_kind, row_owner, row_rhs = np.zeros(cr + 1, np.int32), np.zeros(cr, np.int32), np.zeros(cr, np.int32), np.zeros(cr, np.float32)
    col_idx = np.zeros(cn, np.int32)
    n_cols, n_rows, nnz, nmax = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    L.msorb_visibility_csr.argtypes = ([C.c_int, C.c_int] + [C.c_void_p] * 3 + [C.c_int] + [C.c_void_p] * 3 + [C.c_int] +
                                       [C.c_void_p] * 2 + [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int] +
                                       [C.c_void_p] * 5 + [C.c_int, C.c_void_p, C.c_int] + [C.c_void_p] * 3)
    call = (0, K, p(arrs["kf_slot_begin"]), p(arrs["slot_point"]), p(arrs["slot_cell"]), P, p(arrs["point_nobs"]), p(arrs["obs_begin"]),
            p(arrs["obs_kf"]), KT, p(arrs["kf_in_window"]), p(arrs["kf_num_mps"]), 100, 0, C.byref(n_cols), p(col_point), cc, C.byref(n_rows),
            p(row_begin), p(row_kind), p(row_owner), p(row_rhs), cr, p(col_idx), cn, C.byref(nnz), p(obj), C.byref(nmax))
    ts = []
    for i in range(45):
        t0 = time.perf_counter()
        rc = L.msorb_visibility_csr(*call)
        if i >= 5:
            ts.append(time.perf_counter() - t0)
        if rc:
            raise RuntimeError("msorb_visibility_csr: %d" % rc)
    out = {"what": "configs[4]: constraint matrix of one sparsification window (30 keyframes x 2000 slots, 64x48 grid, 100 outside "
                   "keyframes) as CSR, msorb_visibility_csr through the C ABI, host arrays in and out",
           "ms_per_window": round(float(np.median(ts)) * 1e3, 4), "slots": S, "observations": int(len(arrs["obs_kf"])),
           "cols": n_cols.value, "rows": n_rows.value, "nnz": nnz.value}
    if cpu:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import orb_oracle
        t0 = time.perf_counter()
        reps = 5
        for _ in range(reps):
            want = orb_oracle.visibility_csr(N=100, **w)
        dt = (time.perf_counter() - t0) / reps
        nr = n_rows.value
        same = (want["n_cols"] == n_cols.value and want["n_rows"] == nr and np.array_equal(want["col_point"], col_point[:n_cols.value]) and
                np.array_equal(want["row_begin"], row_begin[:nr + 1]) and np.array_equal(want["col_idx"], col_idx[:nnz.value]))
        self_check(same, "sparsification: msorb_visibility_csr differs from the CPU oracle")
        out["cpu_baseline"] = {"ms_per_window": round(dt * 1e3, 4), "cores": 1, "kind": "port",
                               "sample": f"oracle/sparsify_oracle.cc on the same window, {reps} repetitions", "gpu_matches_cpu": bool(same)}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--pairs", type=int, default=128, help="stereo pairs per step per GPU-pair group")
    ap.add_argument("--cpu-pairs", type=int, default=160, help="stereo pairs in the CPU baseline sample (0 = skip)")
    ap.add_argument("--unique-pairs", type=int, default=8, help="distinct synthetic stereo pairs tiled to the batch (<= --pairs)")
    ap.add_argument("--lean", action="store_true",
                    help="profiling aid: only the extraction loop (no Hamming / stereo / tracking / per-frame / host-fed / sparsification "
                         "legs, no CPU baseline), so that a kernel trace or a counter pass holds nothing else")
    ap.add_argument("--no-pmc", action="store_true", help="do not collect roofline.traffic live with rocprofv3 (child passes of this script)")
    ap.add_argument("--isolated", action="store_true",
                    help="profiling aid: no sub-batch / blur overlap anywhere, so every kernel launch covers the whole "
                         "batch and runs alone (rocprofv3 per-kernel durations and PMC traffic are then per-launch clean)")
    args = ap.parse_args()
    if args.lean:
        args.cpu_pairs = 0

    import torch
    import msorb
    from msorb import synth

    cfg = synth.KITTI
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs WORLD_SIZE={args.gpus} (launch with torch.distributed.run)")
    # test aid: MSORB_DIST_BACKEND=gloo lets the N>1 path run with several ranks on ONE GPU (ranks share cuda:0, the
    # exchange goes through gloo) — how the stereo-split code path is exercised on the 1-GPU development box
    backend = os.environ.get("MSORB_DIST_BACKEND", "nccl")
    if backend != "nccl":
        local = local % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    B = args.pairs
    # images of this rank: N=1 -> L,R interleaved (2B images); N>1 -> one eye of 2B pairs (2B images): fixed per-GPU work
    group, eye = (rank // 2, rank % 2) if world > 1 else (0, None)
    uniq = min(B, args.unique_pairs)  # distinct synthetic pairs, tiled to the batch size (content repeats, bytes do not alias)
    base = synth.stereo_batch(uniq, cfg["rows"], cfg["cols"], seed0=1000 * group)
    pitch = (cfg["cols"] + 63) // 64 * 64

    def resident(host_imgs):
        # device-resident input batch with a 64-byte row pitch (what hipMemcpy2D / a camera DMA would produce);
        # the API accepts any stride, aligned rows take the fast kernel variants
        st = torch.zeros((host_imgs.shape[0], cfg["rows"], pitch), dtype=torch.uint8, device=dev)
        v = st[:, :, :cfg["cols"]]
        v.copy_(torch.from_numpy(np.ascontiguousarray(host_imgs)).to(dev))
        return v

    other_images = None
    half = B   # N>1: images of a rank whose stereo pairs it associates itself (the first `half` of its 2B images)
    if world == 1:
        host = np.concatenate([base] * (B // uniq + 1))[:2 * B]
    else:
        # both ranks of a pair hold the same 2B stereo pairs, one eye each.  The association (Frame::ComputeStereoMatches) is
        # split evenly: the left-eye rank joins pairs [0, B), the right-eye rank pairs [B, 2B) — so the right-eye rank keeps its
        # images in the order [B, 2B) + [0, B): every rank joins the pairs of its FIRST B images and ships the features of its
        # LAST B images to its partner (msorb/stereo_split.swap_halves_async).  A rank also holds the other eye's images of
        # the pairs it joins (the capture DMA delivers both eyes): it rebuilds that pyramid for the stereo SAD locally
        # (msorb_pyramid_batch, 0.09 ms per 128 images) instead of pulling 1.5 MB per frame over xGMI; only the 60 B per
        # keypoint cross the link (BASELINE configs[3]: "gather of keypoints/descriptors")
        def eye_images(e):
            one = base[e::2]
            return np.concatenate([one] * (2 * B // uniq + 1))[:2 * B]
        own, oth = eye_images(eye), eye_images(1 - eye)
        if eye == 1:
            own = np.concatenate([own[B:], own[:B]])
            oth = np.concatenate([oth[B:], oth[:B]])
        host = own
        other_images = resident(oth[:half])
    n_img = host.shape[0]
    images = resident(host)

    def make_ex():
        e = msorb.ORBextractor(cfg["nfeatures"], cfg["scale"], cfg["nlevels"], cfg["ini_th"], cfg["min_th"], device=local)
        if args.isolated:
            e.set_overlap(1, False)
        return e

    ex = make_ex()
    cap = ex.capacity
    d_kps = torch.empty((n_img, cap, 28), dtype=torch.uint8, device=dev)
    d_desc = torch.empty((n_img, cap, 32), dtype=torch.uint8, device=dev)
    from msorb import stereo_split
    # N>1: two send/receive blocks used alternately, so that the exchange of step k (RCCL, torch's communication stream)
    # overlaps the extraction of step k+1 (the library's own streams) without the two touching the same memory.  The
    # left-eye rank alternates two extractor handles the same way: the stereo association of step k (Frame.cc:743-913) runs
    # when the right eye's features of step k have arrived — two steps later, just before block and handle are reused —
    # and needs the left pyramid of step k intact.
    mine = theirs = None
    pending = [[], []]
    exs = [ex, ex]
    ex_rp = None
    assoc = {"ms": 0.0, "n": 0, "matched": 0}
    if world > 1:
        mine = [stereo_split.FeatureBlock(n_img, cap, dev) for _ in range(2)]
        theirs = [stereo_split.FeatureBlock(half, cap, dev) for _ in range(2)]
        exs = [ex, make_ex()]
        if not (args.isolated or os.environ.get("MSORB_BENCH_SYNC")):
            for e in exs:
                e.set_overlap(1, True)   # two batches in flight, one sub-batch each (see the N = 1 loop below)
        ex_rp = make_ex()   # pyramid-only handle for the other eye's images of the pairs this rank joins
    step_no = [0]
    fence_kp = [0]
    filled = [False, False]
    last_ex = [ex]
    all_ex = [ex] if exs[1] is ex else [exs[0], exs[1]]

    def associate(b):
        """Frame::ComputeStereoMatches for the pairs of block b this rank joins (its exchange has completed): own features of
        the first `half` images + the partner's features of the same pairs."""
        if world == 1 or not filled[b]:
            return
        ex_rp.pyramid_batch(other_images)
        own = (mine[b].counts[:half], mine[b].kps[:half], mine[b].desc[:half])
        got = (theirs[b].counts, theirs[b].kps, theirs[b].desc)
        if eye == 0:
            d_ur, _, _, ms = msorb.stereo_matches_split(exs[b], ex_rp, *own, *got, KITTI_MB, KITTI_MBF)
        else:
            d_ur, _, _, ms = msorb.stereo_matches_split(ex_rp, exs[b], *got, *own, KITTI_MB, KITTI_MBF)
        assoc["ms"] += ms
        assoc["n"] += 1
        assoc["last"] = d_ur
        filled[b] = False

    # N=1: two batches in flight.  A synchronous msorb_extract_batch per step leaves the GPU under-used across step boundaries
    # (the last sub-batch's quadtree / descriptors and the next step's small pyramid levels run alone: ~25 % of a step in the
    # kernel timeline); with msorb_extract_batch_submit / _wait on two alternating handles, step k + 1 is enqueued before step k is
    # waited for — what a capture pipeline that always has the next batch ready does.  Every step still runs the full chain
    # on its own batch buffers; the timed region is bracketed by fence() on both sides.  (--isolated / the stage timings use
    # the synchronous call.)
    pipelined = world == 1 and not args.isolated and not os.environ.get("MSORB_BENCH_SYNC")
    mode = {"pipelined": pipelined, "sync_nccl": bool(args.isolated or os.environ.get("MSORB_BENCH_SYNC"))}
    submitted = [False, False]
    depth = int(os.environ.get("MSORB_BENCH_DEPTH", "2")) if pipelined else 1   # batches in flight
    exp = [ex] + [make_ex() for _ in range(depth - 1)]
    all_ex.extend(exp[1:])
    if pipelined:
        # one sub-batch per handle: the concurrency comes from the two batches (measured: 1.35 ms per step against 1.51 with two
        # sub-batches per handle — six streams on four hardware queues — and 1.44 for the one-batch-at-a-time loop)
        for e in exp:
            e.set_overlap(1, True)
    outs = [(d_kps, d_desc)] + [(torch.empty_like(d_kps), torch.empty_like(d_desc)) for _ in range(depth - 1)]
    inflight = []
    refill = [True]   # from a fence until `depth` batches are in flight again
    # Phase between the two batches in flight: started together they stay in lock-step (both finish, and are resubmitted,
    # together: FAST beside FAST, descriptors beside descriptors); half a step apart, one batch's pyramid / FAST (VALU issue)
    # runs beside the other's quadtree (latency) and descriptors (line fills): 1.147 instead of 1.184 ms per step (round 4; a
    # capture pipeline whose batches arrive evenly spaced is in this state by itself).  The offset is half of the step time
    # measured over the warm-up steps (MSORB_BENCH_STAGGER_US overrides; 0 = lock-step), applied once after every fence, inside
    # the timed region.
    stagger = [float(os.environ["MSORB_BENCH_STAGGER_US"]) * 1e-6 if "MSORB_BENCH_STAGGER_US" in os.environ else None]
    if not pipelined or args.steps < 8:   # (a handful of steps cannot pay for the half step the offset costs once)
        stagger[0] = 0.0

    def drain():
        total = 0
        refill[0] = True
        while inflight:
            counts, _, _, _ = inflight.pop(0).extract_batch_wait()
            total += int(counts.sum())
        return total

    def step():
        if world == 1:
            if not mode["pipelined"]:
                counts, mono, _, _ = ex.extract_batch(images, (0, 0), out=(d_kps, d_desc))
                last_ex[0] = ex
                return int(counts.sum())
            k = step_no[0] % depth
            step_no[0] += 1
            done = 0
            if len(inflight) == depth:  # handle k still holds the batch submitted `depth` steps ago
                counts, _, _, _ = inflight.pop(0).extract_batch_wait()
                done = int(counts.sum())
            exp[k].extract_batch_submit(images, (0, 0), out=outs[k])
            inflight.append(exp[k])
            if len(inflight) == depth:
                refill[0] = False
            last_ex[0] = exp[k]
            if 0 < len(inflight) < depth and refill[0] and stagger[0]:
                time.sleep(stagger[0])   # first batch after a fence: hold the second one back (see `stagger` above)
            return done
        b = step_no[0] & 1
        step_no[0] += 1
        stereo_split.finish(pending[b])          # the block's previous exchange (started one step ago) must be over
        pending[b] = []
        associate(b)                              # ... and its pairs are joined before block / handle are reused
        if mode["sync_nccl"]:                     # stage timings: one batch at a time
            last_ex[0] = exs[b]
            counts, mono, _, _ = exs[b].extract_batch(images, (0, 0), out=(mine[b].kps, mine[b].desc))
            ship(b, counts)
            return int(counts.sum())
        # two batches in flight, as at N = 1: this step's extraction is enqueued (the extractor writes straight into the send
        # block), then the previous step's is waited for and its feature block goes on the wire
        exs[b].extract_batch_submit(images, (0, 0), out=(mine[b].kps, mine[b].desc))
        submitted[b] = True
        return collect(b ^ 1)

    def ship(b, counts):
        """stereo split: swap the feature blocks of block b with the partner (replaces the join of Frame.cc:122-125)"""
        mine[b].counts.copy_(torch.from_numpy(counts))
        pending[b] = stereo_split.swap_halves_async(dist, rank, world, mine[b], theirs[b])
        filled[b] = bool(pending[b])

    def collect(b):
        if not submitted[b]:
            return 0
        counts, _, _, _ = exs[b].extract_batch_wait()
        submitted[b] = False
        last_ex[0] = exs[b]
        ship(b, counts)
        return int(counts.sum())

    def fence():
        if mode["pipelined"]:
            fence_kp[0] += drain()
        if world > 1:
            for b in range(2):
                fence_kp[0] += collect(b)
            for b in range(2):
                stereo_split.finish(pending[b])
                pending[b] = []
                associate(b)
            dist.barrier()
        torch.cuda.synchronize()

    # Order of the untimed work: the Hamming, stereo-association and per-kernel stage measurements run BEFORE the warm-up steps,
    # the CPU legs after the timed region.  A GPU that comes out of idle runs its first ~25 ms of work 5-15 % slower (clock
    # ramp: tools/ramp.py prints the per-step times), so the stage measurement discards its first 20 steps — and the timed
    # region that follows starts on a GPU at its working clocks even with `--steps 20 --warmup 5`.
    # second half of the metric: Gpairs/s of the brute-force Hamming match (left-eye descriptors of every pair
    # against the right-eye descriptors of the same pair, dense top-2), on the descriptors just extracted
    hamming = None
    aux = world == 1 and not args.lean
    if aux:
        counts_h, _, _, _ = ex.extract_batch(images, (0, 0), out=(d_kps, d_desc))
        dq = d_desc[0::2].contiguous()
        dtr = d_desc[1::2].contiguous()
        nq = torch.from_numpy(np.ascontiguousarray(counts_h[0::2])).to(dev)
        nt = torch.from_numpy(np.ascontiguousarray(counts_h[1::2])).to(dev)
        msorb.hamming_dense_top2_batch(dq, dtr, nq, nt, repeats=2, device=local)   # warm-up
        reps = 60
        _, _, _, ms = msorb.hamming_dense_top2_batch(dq, dtr, nq, nt, repeats=reps, device=local)
        pairs = int((counts_h[0::2].astype(np.int64) * counts_h[1::2].astype(np.int64)).sum())
        # The kernel runs on the matrix cores (matcher.hip dense_top2_mfma_kernel): +-32 int8 encoding, 8 x v_mfma_i32_32x32x32_i8 per
        # 32 x 32 pairs = 512 int8 operations per pair.  Ceiling = the i8 MFMA rate this chip sustains with nothing else running
        # (4.3 POPS: 37.5 cycles@2.4GHz per instruction and SIMD, tools/mfma_rate.hip; docs/DESIGN_rounds1-3.md section 4).
        # For reference the integer-VALU formulation: 8 x (v_xor + accumulating v_bcnt) per 64 pairs at 7.91 cycles@2.4GHz per
        # instruction PAIR (tools/valu_ubench2.hip, round 4: in a stream that mixes the two classes a fast-class instruction costs
        # as much as a slow one, whether alternating or in runs of 16 — 2.5 cycles hold in pure fast-class streams only), + 3
        # slow-class instructions of top-2 bookkeeping at 4.2.
        g = pairs * reps / (ms * 1e-3) / 1e9
        mfma_pops = 4.3e15
        ceil_mfma = mfma_pops / 512 / 1e9
        ceil_valu = 1024 * 2.4e9 * 64 / (8 * 7.91) / 1e9
        ceil_valu_top2 = 1024 * 2.4e9 * 64 / (8 * 7.91 + 3 * 4.2) / 1e9
        # the north_star's own formulation (xor + __builtin_popcount per pair, no MFMA) measured beside it on the same descriptors
        msorb.hamming_dense_top2_batch(dq, dtr, nq, nt, repeats=2, device=local, formulation=msorb.DENSE_POPCOUNT)
        bi_v, bd_v, sd_v, ms_v = msorb.hamming_dense_top2_batch(dq, dtr, nq, nt, repeats=reps, device=local,
                                                                formulation=msorb.DENSE_POPCOUNT)
        bi_m, bd_m, sd_m, _ = msorb.hamming_dense_top2_batch(dq, dtr, nq, nt, repeats=1, device=local)
        # rows >= nq[f] of a frame are never written by either kernel: only the valid rows are results
        live = torch.arange(dq.shape[1], device=dev)[None, :] < nq[:, None]
        same_kernels = bool(torch.equal(bi_v[live], bi_m[live]) and torch.equal(bd_v[live], bd_m[live]) and
                            torch.equal(sd_v[live], sd_m[live]))
        self_check(same_kernels, "hamming_match: the popcount and the MFMA kernel disagree on a valid row")
        popcount_out = (bi_v, bd_v, sd_v)
        g_v = pairs * reps / (ms_v * 1e-3) / 1e9
        hamming = {"gpairs_per_s": round(g, 2), "pairs_per_launch": pairs,
                   "ms_per_launch": round(ms / reps, 4), "kernel": "dense_top2_mfma_kernel (v_mfma_i32_32x32x32_i8)",
                   "formulation": "MSORB_DENSE_MATRIX_CORES: Hamming distance as an int8 dot product on the matrix cores — exact, identical "
                                  "results, but NOT the north_star's formulation ('no MFMA'); the conformant figure is popcount_kernel below",
                   "ceiling_gpairs_per_s": round(ceil_mfma, 1), "frac": round(g / ceil_mfma, 3),
                   "ceiling_note": "i8 MFMA rate measured on this chip (4.3 POPS) / 512 operations per pair",
                   "valu_formulation_ceiling_gpairs_per_s": round(ceil_valu, 1),
                   "valu_formulation_ceiling_with_top2_gpairs_per_s": round(ceil_valu_top2, 1),
                   "bound": "matrix-core issue (MFMA) with the top-2 bookkeeping (v_med3 + v_min per pair) interleaved under it; "
                            "not HBM: (Q+T)*32 B per frame are reused Q*T times",
                   "popcount_kernel": {"what": "the north_star's formulation: dense_top2_kernel<2, 4>, v_xor + accumulating v_bcnt per "
                                               "dword, no MFMA (formulation MSORB_DENSE_POPCOUNT); same inputs, same launch count",
                                       "gpairs_per_s": round(g_v, 2), "ms_per_launch": round(ms_v / reps, 4),
                                       "frac_of_valu_ceiling_with_top2": round(g_v / ceil_valu_top2, 3),
                                       "ceiling_note": "mixed-stream VALU issue rate measured on this chip (tools/valu_ubench2.hip: "
                                                       "v_xor + v_bcnt = 7.91 cycles@2.4GHz per pair of instructions); PMC of this "
                                                       "kernel in profiles/round4_dense_popcount_pmc.txt",
                                       "identical_results": same_kernels}}
    # third: Frame::ComputeStereoMatches for the whole batch, device resident (pair p = images 2p / 2p+1), median
    # rejection included; the outputs of the extraction above are its inputs
    stereo = None
    if aux:
        msorb.stereo_matches_batch(ex, counts_h, d_kps, d_desc, KITTI_MB, KITTI_MBF)
        sms = [msorb.stereo_matches_batch(ex, counts_h, d_kps, d_desc, KITTI_MB, KITTI_MBF)[3] for _ in range(15)]
        d_ur, _, _, _ = msorb.stereo_matches_batch(ex, counts_h, d_kps, d_desc, KITTI_MB, KITTI_MBF)
        m = float(np.median(sms))
        n_left = int(counts_h[0::2].sum())
        stereo = {"pairs": int(len(counts_h) // 2), "left_keypoints": n_left, "matched": int((d_ur > 0).sum().item()),
                  "ms_per_batch": round(m, 4), "mkeypoints_per_s": round(n_left / (m * 1e-3) / 1e6, 2),
                  "kernels": "stereo_rowtable_kernel + stereo_match_quad_kernel (four left keypoints per wave) + stereo_median_kernel"}

    # fourth: BASELINE configs[2] — the front-end of one tracking frame as a device-resident chain (csrc/track.hip): extraction of
    # both eyes + ComputeStereoMatches + AssignFeaturesToGrid + isInFrustum + the window search of SearchByProjection
    tracking = None
    if aux:
        tracking = tracking_loop_leg(msorb, synth, torch, ex, cfg, base, counts_h, d_kps, d_desc, d_ur, dev, local, args)

    per_frame = host_fed = None
    if aux:
        per_frame = per_frame_leg(msorb, ex, base[0], base[1])

    # per-kernel roofline: the same step with every kernel alone on the GPU (1 sub-batch, blur on the main stream),
    # HIP events on the launching stream, 20 recorded steps (after 20 discarded ones) outside the timed region
    iso_steps, iso_discard = 20, 20
    for e in all_ex:
        e.set_profiling(True)
    saved_mode = dict(mode)
    stage_acc = {k: [] for k in msorb.STAGES}
    mode["pipelined"] = False   # the stage timings use the synchronous call on one handle
    mode["sync_nccl"] = True
    for e in all_ex:
        e.set_overlap(1, False)
    for _ in range(iso_discard):   # not recorded: lazy allocations, and the GPU's clocks ramp for ~25 ms after idle
        step()
    for _ in range(iso_steps):
        step()
        for k, v in last_ex[0].stage_ms().items():
            stage_acc[k].append(v)
    if not args.isolated:
        for e in all_ex:
            e.set_overlap(1 if pipelined else 2, True)
    fence()
    mode.update(saved_mode)



    t_w = time.perf_counter()
    for _ in range(args.warmup):
        step()
    if pipelined and stagger[0] is None:
        fence()
        stagger[0] = (time.perf_counter() - t_w) / args.warmup / depth if args.warmup >= 2 else 1200e-6 / depth
        stagger[0] = min(max(stagger[0], 200e-6), 2e-3)
    # timed region: the production shape (2 sub-batches in flight, blur on a second stream), stage events on
    for e in all_ex:
        e.set_profiling(True)
    overlapped_acc = {k: 0.0 for k in msorb.STAGES}
    fence()
    assoc.update(ms=0.0, n=0)
    fence_kp[0] = 0
    t0 = time.perf_counter()
    kp_total = 0
    for _ in range(args.steps):
        kp_total += step()
        if not pipelined and (world == 1 or mode["sync_nccl"]):
            for k, v in last_ex[0].stage_ms().items():
                overlapped_acc[k] += v
    fence()
    dt = time.perf_counter() - t0
    kp_total += fence_kp[0]
    join = dict(assoc)   # association statistics of the timed region only

    dt_rank, kp_total_rank = dt, kp_total
    if world > 1:
        t = torch.tensor([dt, float(kp_total)], dtype=torch.float64, device=dev)
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        dt, kp_total = float(tmax[0]), int(t[1])

    if aux and not args.isolated:
        hf_ex = [ex, make_ex()]
        host_fed = host_fed_leg(msorb, torch, hf_ex, host, dev, cfg, pitch)
        all_ex.append(hf_ex[1])

    validation = None
    if world > 1:
        validation = split_self_validation(msorb, torch, dist, stereo_split, rank, world, eye, half, exs[0], ex_rp, make_ex, images,
                                           other_images, mine[0], theirs[0], kp_total_rank / dt_rank if dt_rank > 0 else 0.0, dev)

    if hamming is not None:
        if args.cpu_pairs > 0 and rank == 0:
            # CPU leg of the matcher on a bounded sample: ORBmatcher::DescriptorDistance brute force (oracle, 1 thread) on
            # the first stereo pair's descriptors; its result also cross-checks the GPU's indices and distances
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            import orb_oracle
            bi_g, bd_g, sd_g, _ = msorb.hamming_dense_top2_batch(dq, dtr, nq, nt, repeats=1, device=local)
            n0, n1 = int(counts_h[0]), int(counts_h[1])
            q0, t0_ = dq[0, :n0].cpu().numpy(), dtr[0, :n1].cpu().numpy()
            tc = time.perf_counter()
            bi_c, bd_c, sd_c = orb_oracle.dense_top2(q0, t0_)
            dtc = time.perf_counter() - tc
            # the same brute force over many frames: one core (16 frames) and every host core (all 128 frames, one frame per task)
            from concurrent.futures import ThreadPoolExecutor
            dq_h, dt_h = dq.cpu().numpy(), dtr.cpu().numpy()
            frames16 = list(range(min(16, dq_h.shape[0])))
            tc1 = time.perf_counter()
            for f_ in frames16:
                orb_oracle.dense_top2(dq_h[f_, :int(counts_h[2 * f_])], dt_h[f_, :int(counts_h[2 * f_ + 1])])
            dt1 = time.perf_counter() - tc1
            pairs1 = sum(int(counts_h[2 * f_]) * int(counts_h[2 * f_ + 1]) for f_ in frames16)
            ncore = min(os.cpu_count() or 1, dq_h.shape[0])
            tca = time.perf_counter()
            with ThreadPoolExecutor(ncore) as pool:
                list(pool.map(lambda f_: orb_oracle.dense_top2(dq_h[f_, :int(counts_h[2 * f_])], dt_h[f_, :int(counts_h[2 * f_ + 1])]),
                              range(dq_h.shape[0])))
            dta = time.perf_counter() - tca
            same = (np.array_equal(bi_c, bi_g[0, :n0].cpu().numpy()) and np.array_equal(bd_c, bd_g[0, :n0].cpu().numpy()) and
                    np.array_equal(sd_c, sd_g[0, :n0].cpu().numpy()))
            same_pop = all(np.array_equal(c, g[0, :n0].cpu().numpy()) for c, g in zip((bi_c, bd_c, sd_c), popcount_out))
            self_check(same, "hamming_match: the MFMA kernel differs from the CPU oracle")
            self_check(same_pop, "hamming_match: the popcount kernel differs from the CPU oracle")
            hamming["cpu_baseline"] = {"gpairs_per_s": round(pairs1 / dt1 / 1e9, 4), "cores": 1, "kind": "port",
                                       "sample": f"{len(frames16)} stereo pairs of ~{n0} x {n1} descriptors, {dt1 * 1e3:.1f} ms "
                                                 "(xor + __builtin_popcountll, -O3 x86-64-v3)",
                                       "all_cores": {"gpairs_per_s": round(pairs / dta / 1e9, 3), "cores": ncore,
                                                     "sample": f"all {dq_h.shape[0]} pairs, one frame per task, {dta * 1e3:.1f} ms"},
                                       "gpu_matches_cpu": bool(same), "popcount_kernel_matches_cpu": bool(same_pop)}

    if rank == 0:
        steps = max(args.steps, 1)
        # median over the recorded steps (a mean is at the mercy of a few slow steps: clock ramps, a neighbour on the host)
        stages = {k: float(np.median(v)) if v else 0.0 for k, v in stage_acc.items()}
        stages_overlapped = {k: v / steps for k, v in overlapped_acc.items()}
        px = level_bytes(cfg)
        # dominant GPU kernel: FAST cells (one launch per step).  Algorithmic bytes per launch:
        # every level pixel read once + 8 B per emitted candidate (not counted: unknown a priori) per image.
        gpu_stages = {k: stages[k] for k in ("pyramid", "fast", "blur", "describe", "compact")}
        dom = max(gpu_stages, key=gpu_stages.get)
        alg_per_image = {
            "fast": sum(px),                                 # read every level once
            "pyramid": sum(px[:-1]) + sum(px[1:]),           # read L0..L6, write L1..L7 (L0 consumed in place)
            "blur": 2 * sum(px),                             # read + write every level
            "describe": cfg["nfeatures"] * (749 + 512 + 32 + 28),
            "compact": 0,
        }[dom]
        alg_bytes = alg_per_image * n_img
        achieved = alg_bytes / (stages[dom] * 1e-3) / 1e9 if stages[dom] > 0 else 0.0
        pf_bytes = (alg_per_image if dom in ("fast", "pyramid") else 0)
        # HBM traffic of the dominant kernel: PMC counters cannot be read from inside this process, so the value is the
        # committed rocprofv3 --pmc measurement of the same command (latest profiles/roundN_pmc_summary.json: separate
        # FETCH_SIZE / WRITE_SIZE passes, per launch, FETCH_SIZE scaled by the calibration kernels of the same run) — only
        # quoted when the batch shape matches
        traffic = None
        kname = {"fast": "fast_cells_kernel<true, GeoSmall>", "pyramid": "pyr_resize_bandreg_kernel<8, 12, 2>", "blur": "gauss7_stream_kernel<35>",
                 "describe": "describe_kernel", "compact": "cand_gather_kernel"}[dom]
        valu_frac = None
        pmc_file = None
        try:
            import glob
            pmc_file = sorted(glob.glob(os.path.join(ROOT, "profiles", "round*_pmc_summary.json")))[-1]
            summ = json.load(open(pmc_file))
            pm = summ["kernels"][kname]
            if B == 128 and world == 1:
                scale = summ.get("fetch_scale", 2.0)
                traffic = int((pm["FETCH_SIZE_KB"] * scale + pm["WRITE_SIZE_KB"]) * 1024) * (7 if dom == "pyramid" else 1)
                valu_frac = pm.get("valu_fraction_of_measured_peak")
        except Exception:
            traffic = None
        traffic_source = ("committed " + os.path.basename(pmc_file)) if (traffic is not None and pmc_file) else None
        if aux and not args.no_pmc and not args.isolated and B == 128 and dom == "fast":
            pm_live = live_pmc("fast_cells_kernel")
            if pm_live:
                scale = 2.0    # FETCH_SIZE reports half of the bytes a coalesced stream reads on gfx950 (profiles/round4_fetch_calib.txt)
                traffic = int((pm_live["FETCH_SIZE"] * scale + pm_live["WRITE_SIZE"]) * 1024)
                valu_frac = round(pm_live["SQ_INSTS_VALU"] * 64 / (stages[dom] * 1e-3) / 51.5e12, 3)
                traffic_source = ("measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_INSTS_VALU, one pass each, over a child "
                                  "`bench.py --lean --isolated` of the same batch")
        out = {
            "metric": "Mkeypoints/s extract+describe, KITTI-00-like stereo 1241x376, 2000 feat/frame",
            "value": round(kp_total / dt / 1e6, 4),
            "unit": "Mkeypoints/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt / steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "metric_notes": {"inputs": "value assumes the images are already in HBM (bench contract); host_fed is the PCIe-inclusive batch "
                                       "rate, per_frame what one frame at a time from host memory costs",
                             "hamming_match": "gpairs_per_s is an int8 matrix-core formulation (a deviation from north_star's 'no MFMA'); "
                                              "hamming_match.popcount_kernel is the north_star's xor + popcount form, same inputs, same run",
                             "two_gpus": "the product's stereo split (msorb_extract_stereo_split) gathers with hipMemcpyPeerAsync inside one "
                                         "process; RCCL (torch.distributed nccl) carries the exchange only in `bench.py --gpus N`",
                             "see": "BASELINE.md section 3"},
            "config": {"workload": "configs[1]: KITTI-00 stereo 1241x376, 2000 feat/frame, pyramid+FAST+rBRIEF",
                       "split_transport": None if world == 1 else "RCCL point-to-point (torch.distributed nccl backend), one rank per GPU",
                       "pairs_per_step_per_gpu": B if world == 1 else B, "images_per_step_per_gpu": n_img,
                       "stagger_us": round((stagger[0] or 0.0) * 1e6, 1),
                       "batches_in_flight": depth if world == 1 else (1 if args.isolated or os.environ.get("MSORB_BENCH_SYNC") else 2),
                       "parallelism": ("1 GPU, both eyes; msorb_extract_batch_submit / _wait on two alternating handles: step k+1 is enqueued "
                                       "before step k is waited for" if pipelined else "1 GPU, both eyes") if world == 1 else f"stereo L/R split over {world} GPUs: one eye per rank; partners swap the keypoints / "
                                                                             "descriptors of half of their images (RCCL send/recv over xGMI) and each joins half of the pairs (stereo association)"},
            "stage_ms_per_step": {k: round(v, 4) for k, v in stages.items()},
            "stage_ms_per_step_note": "each stage's kernels alone on the GPU (20 extra steps after 20 discarded ones, median, overlap off, HIP events on the "
                                      "launching stream); 'select' = device quadtree + output layout",
            "stage_ms_per_step_overlapped": None if (pipelined or (world > 1 and not args.isolated and not os.environ.get("MSORB_BENCH_SYNC"))) else {k: round(v, 4) for k, v in stages_overlapped.items()},
            "stage_ms_per_step_overlapped_note": "timed region: sum over the 2 concurrent sub-batches of each stage's event "
                                                 "interval (intervals overlap, so the sum exceeds ms_per_step); not recorded when "
                                                 "two batches are in flight (MSORB_BENCH_SYNC=1 for the one-batch-at-a-time loop)",
            "roofline": {"bound": "hbm", "kernel": {"fast": "fast_cells_kernel", "pyramid": "pyr_resize_bandreg_kernel (x7)",
                                                    "blur": "gauss7_kernel (x8)", "describe": "describe_kernel",
                                                    "compact": "cand_*"}[dom],
                         "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "traffic_source": traffic_source,
                         "traffic_note": "HBM bytes per launch from rocprofv3 --pmc: FETCH_SIZE x 2 + WRITE_SIZE; FETCH_SIZE reports half of "
                                         "the bytes a coalesced stream reads on gfx950 (tools/fetch_calib.hip, profiles/round4_fetch_calib.txt), "
                                         "WRITE_SIZE is exact",
                         "algorithmic_bytes_per_launch": int(alg_bytes),
                         "valu_fraction_of_measured_peak": valu_frac,
                         "valu_note": "what actually bounds this kernel: 64 x SQ_INSTS_VALU / duration against the 51.5 T lane-ops/s "
                                      "the VALUs sustain (103 TFLOP/s v_fma_f32); same source as traffic"},
            "pyramid_fast_gbs": round((sum(px[:-1]) + sum(px[1:]) + sum(px)) * n_img /
                                      ((stages["pyramid"] + stages["fast"]) * 1e-3) / 1e9, 2),
        }
        out["hamming_match"] = hamming
        out["stereo_match"] = stereo
        if tracking is not None and args.cpu_pairs > 0:
            tracking_cpu_leg(tracking, msorb, os.path.join(ROOT, "oracle"))
        if tracking is not None:
            tracking.pop("_cpu", None)
            tracking.pop("_cpu_mm", None)
        if tracking is not None:
            tracking["reference_keyframe"] = reference_keyframe_leg(msorb, args.cpu_pairs > 0)
        out["tracking_loop"] = tracking
        if aux:
            out["sparsification"] = sparsification_leg(msorb, args.cpu_pairs > 0)
        if world > 1:
            out["stereo_join"] = {
                "what": "every rank joins half of its pair group's stereo pairs inside the timed region, once per step: the other "
                        "eye's pyramid of those pairs rebuilt from resident images (msorb_pyramid_batch) + "
                        "Frame::ComputeStereoMatches on its own features and the features gathered from its partner "
                        "(msorb_stereo_matches_split); figures of rank 0",
                "pairs_per_join": half,
                "joins": join["n"], "kernel_ms_per_join": round(join["ms"] / max(join["n"], 1), 4),
                "matched_last": int((join["last"] > 0).sum().item()) if "last" in join else None,
                "gathered_bytes_per_step_and_rank": theirs[0].nbytes() if theirs else None}
        out["per_frame"] = per_frame
        out["host_fed"] = host_fed
        if validation is not None:
            out["split_validation"] = validation
        if world == 1 and args.cpu_pairs > 0:
            out["cpu_baseline"] = cpu_baseline(cfg, args.cpu_pairs, 5000)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    for e in all_ex + ([ex_rp] if ex_rp is not None else []):
        e.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
