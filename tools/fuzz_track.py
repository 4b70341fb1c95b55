#!/usr/bin/env python3
"""Random sweep of the tracking front-end chain (csrc/track.hip) against the CPU oracle, beyond the fixed cases of
tests/test_track_gpu.py: image geometries (KITTI / EuRoC / 4Seasons / odd sizes), feature counts, camera intrinsics and poses,
search radii, far-point gates, frames that already hold map points, local maps from empty to 5x the keypoints.  Every case checks
msorb_track_frontend (one call) and msorb_extract_stereo_frame + msorb_search_local_points (two calls) against
oracle extraction + isInFrustum + SearchByProjection, and msorb_track_frontend_motion + msorb_search_last_frame (a14, projection
on the device, forward / backward bands, retry on the resident table, frame already holding points) against the oracle's
projection + SearchByProjection(Current, Last).  usage (GPU box): python tools/fuzz_track.py [n_cases] [seed0]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "ms-slam_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import msorb
import orb_oracle
from msorb import synth

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
GEOS = [(376, 1241), (480, 752), (400, 800), (333, 517), (240, 320)]
bad = 0
for case in range(n_cases):
    rng = np.random.Generator(np.random.PCG64(seed0 + case))
    rows, cols = GEOS[rng.integers(0, len(GEOS))]
    nfeat = int(rng.choice([300, 1000, 2000, 3500]))
    L, R = synth.stereo_pair(1000 + seed0 + case, rows, cols, max_disp=float(rng.uniform(10, 60)))
    fx = float(rng.uniform(350, 900))
    cam = dict(fx=fx, fy=fx * float(rng.uniform(0.95, 1.05)), cx=cols / 2 + float(rng.uniform(-20, 20)), cy=rows / 2 + float(rng.uniform(-10, 10)),
               mbf=float(rng.uniform(100, 500)))
    mbf, mb = cam["mbf"], cam["mbf"] / cam["fx"]
    bounds = (0.0, float(cols), 0.0, float(rows))
    ex = msorb.ORBextractor(nfeat, 1.2, 8, 20, 7)
    ref = [orb_oracle.OracleExtractor(nfeat, 1.2, 8, 20, 7) for _ in range(2)]
    scale = ex.GetScaleFactors()
    w = rng.normal(scale=0.03, size=3)
    ang = np.linalg.norm(w) + 1e-12
    k = w / ang
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    Rm = (np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * K @ K).astype(np.float32)
    t = rng.normal(scale=0.4, size=3).astype(np.float32)
    # oracle side: extraction, stereo association
    _, okl, odl = ref[0](L)
    _, okr, odr = ref[1](R)
    pl, pr = [ref[0].level(l) for l in range(8)], [ref[1].level(l) for l in range(8)]
    tb = ref[0].tables()
    our, odp, _ = orb_oracle.compute_stereo_matches(okl, odl, okr, odr, pl, pr, tb["scale"], tb["inv_scale"], mb, mbf)
    M = int(rng.choice([0, 50, 1500, 4096, 5 * max(len(okl), 1)]))
    mp = synth.local_map(7000 + seed0 + case, okl, odl, our * 0 + odp if False else odp, scale, M, pose=(Rm, t), cam=dict(cam), bounds=bounds,
                         copy_frac=float(rng.uniform(0.3, 0.95)), obs_zero_frac=float(rng.uniform(0, 0.5)))
    fr = msorb.Frustum.make(mp["Rcw"], mp["tcw"], mp["Ow"], cam["fx"], cam["fy"], cam["cx"], cam["cy"], bounds, mbf, float(np.log(np.float32(1.2))), 8)
    th = float(rng.choice([1.0, 1.0, 3.0, 5.0]))
    far = bool(rng.random() < 0.3)
    th_far = float(rng.uniform(10, 60))
    nnratio = float(rng.choice([0.8, 0.9, 0.6]))

    def oracle_search(frame_mp):
        rf = orb_oracle.OracleFrame(okl, odl, our, bounds, scale)
        r = orb_oracle.is_in_frustum(fr, mp["pos_w"], mp["normal"], mp["max_distance"], mp["min_distance"], 0.5)
        tab = dict(track_in_view=(r["track_in_view"].astype(bool) & mp["visit"].astype(bool)).astype(np.uint8), bad=mp["bad"],
                   sparsified=mp["sparsified"], proj_x=r["proj_x"], proj_y=r["proj_y"], proj_xr=r["proj_xr"], track_depth=r["track_depth"],
                   level=r["level"], view_cos=r["view_cos"], desc=mp["desc"], obs=mp["obs"])
        return rf.SearchByProjection_mps(tab, frame_mp, th, far, th_far, nnratio)
    ok = True
    parts = []
    try:
        # one call
        f, st, frame_mp, nm, out, rounds = msorb.track_frontend(ex, L, R, mb, mbf, fr, mp, th, far, th_far, nnratio, bounds=bounds)
        ok &= np.array_equal(st[0].view(np.uint8), okl.view(np.uint8)) and np.array_equal(st[1], odl)
        ok &= np.array_equal(st[4].view(np.uint32), our.view(np.uint32)) and np.array_equal(st[5].view(np.uint32), odp.view(np.uint32))
        want = np.full(len(okl), -1, np.int32)
        rnm = oracle_search(want)
        ok &= nm == rnm and np.array_equal(frame_mp, want)
        parts = ["one_call" if ok else "ONE_CALL"]
        f.close()
        # two calls, frame already holding map points
        f2, st2 = msorb.extract_stereo_frame(ex, L, R, mb, mbf, bounds=bounds)
        init = (np.where(rng.random(len(okl)) < 0.3, rng.integers(0, max(M, 1), len(okl)), -1).astype(np.int32) if M else np.full(len(okl), -1, np.int32))
        a, b = init.copy(), init.copy()
        nm2, _ = msorb.search_local_points(f2, fr, mp, a, th, far, th_far, nnratio)
        rnm2 = oracle_search(b)
        ok2 = nm2 == rnm2 and np.array_equal(a, b)
        parts.append("two_calls" if ok2 else "TWO_CALLS")
        ok &= ok2
        f2.close()
        # TrackWithMotionModel's search with the projection on the device (a14): fused with the extraction, then the retry at
        # 2 * th on the resident table with the frame already holding points, both against the oracle composition
        last, q, tq, _, _ = synth.last_frame(9000 + seed0 + case, okl, odl, odp, point_frac=float(rng.uniform(0.3, 1.0)),
                                              obs_zero_frac=float(rng.uniform(0, 0.5)), pixel_sigma=float(rng.uniform(0.5, 6.0)),
                                              behind_frac=float(rng.uniform(0, 0.1)), cam=cam) if len(okl) else (None,) * 5
        if last is not None:
            fwd, bwd = bool(rng.random() < 0.25), False
            if not fwd:
                bwd = bool(rng.random() < 0.25)
            th14 = float(rng.choice([7.0, 15.0, 3.0]))
            ori = bool(rng.random() < 0.8)
            mm = msorb.MotionModel.make(q, tq, cam["fx"], cam["fy"], cam["cx"], cam["cy"], cam["mbf"], fwd, bwd)
            omm = orb_oracle.MotionModel()
            omm.q[:] = [float(v) for v in q]
            omm.t[:] = [float(v) for v in tq]
            omm.fx, omm.fy, omm.cx, omm.cy, omm.mbf = cam["fx"], cam["fy"], cam["cx"], cam["cy"], cam["mbf"]
            valid, pu, pv, pur = orb_oracle.project_last_frame(omm, bounds, last["has_point"], last["pos_w"])
            nl = len(valid)

            def oracle14(cur, obs, thx):
                rf = orb_oracle.OracleFrame(okl, odl, our, bounds, scale)
                tab = dict(valid=valid, u=pu, v=pv, ur=pur, octave=last["octave"], angle=last["angle"], desc=last["desc"],
                           mp=np.arange(nl, dtype=np.int32), obs=obs)
                return rf.SearchByProjection_frames(tab, cur, thx, fwd, bwd, ori)
            f3, st3, cur, nm3 = msorb.track_frontend_motion(ex, L, R, mb, mbf, mm, last, last["obs"], th14, ori, bounds=bounds)
            want = np.full(len(okl), -1, np.int32)
            ok3 = nm3 == oracle14(want, last["obs"], th14) and np.array_equal(cur, want)
            parts.append("motion" if ok3 else "MOTION")
            ok &= ok3
            held = rng.random(len(okl)) < 0.25
            extra = np.where(rng.random(len(okl)) < 0.4, 0, rng.integers(1, 9, len(okl))).astype(np.int32)
            obs2 = np.concatenate([last["obs"], extra])
            c0 = np.where(held, nl + np.arange(len(okl)), -1).astype(np.int32)
            a, b = c0.copy(), c0.copy()
            nm4, proj = msorb.search_last_frame(f3, mm, obs2, a, 2 * th14, ori, want_projection=True)
            ok4 = nm4 == oracle14(b, obs2, 2 * th14) and np.array_equal(a, b)
            ok5 = all(np.array_equal(proj[k].view(np.uint8), w.view(np.uint8)) for k, w in (("valid", valid), ("u", pu), ("v", pv), ("ur", pur)))
            parts += ["retry" if ok4 else "RETRY", "projection" if ok5 else "PROJECTION"]
            ok &= ok4 and ok5
            f3.close()
    except Exception as e:   # noqa: BLE001
        print("case", case, "exception", e)
        ok = False
    ex.close()
    print(f"case {case}: {rows}x{cols} nfeat {nfeat} kps {len(okl)} M {M} th {th} far {far} -> matches {nm if ok else '?'} rounds {rounds if ok else '?'} {'ok' if ok else 'MISMATCH ' + ' '.join(parts)}",
          flush=True)
    bad += not ok
print("cases", n_cases, "mismatches", bad)
sys.exit(1 if bad else 0)
