#!/usr/bin/env python3
"""One tracking frame through the C ABI, repeated: msorb_frame_set_last_points + msorb_track_frontend_motion (Frame::Frame +
TrackWithMotionModel's search) and msorb_search_local_points (TrackLocalMap's SearchLocalPoints) on bench.py's synthetic
KITTI-like pair.  Prints median wall times; under rocprofv3 (tools/frame_trace.sh) the last frames give the timeline."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "ms-slam_amd")]
import msorb
from msorb import synth

MBF = 386.1448
MB = MBF / 718.856


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    cfg = synth.KITTI
    L, R = synth.stereo_pair(0, cfg["rows"], cfg["cols"])
    ex = msorb.ORBextractor(cfg["nfeatures"], cfg["scale"], cfg["nlevels"], cfg["ini_th"], cfg["min_th"])
    kl, dl, kr, dr, ur, dp, oob = ex.extract_stereo(L, R, MB, MBF)
    scale = ex.GetScaleFactors()
    cam = synth.KITTI_CAM
    bounds = (0.0, float(cfg["cols"]), 0.0, float(cfg["rows"]))
    last, q, t, fw, bw = synth.last_frame(9500, kl, dl, dp)
    mm = msorb.MotionModel.make(q, t, cam["fx"], cam["fy"], cam["cx"], cam["cy"], cam["mbf"], fw, bw)
    mp = synth.local_map(9000, kl, dl, dp, scale, 4096)
    fr = msorb.Frustum.make(mp["Rcw"], mp["tcw"], mp["Ow"], cam["fx"], cam["fy"], cam["cx"], cam["cy"], bounds, cam["mbf"],
                            float(np.log(np.float32(cfg["scale"]))), cfg["nlevels"])
    run = msorb.MotionFrontendRunner(ex, L, R, MB, MBF, mm, last, last["obs"], 7.0)
    run.attach_local_points(fr, mp, 1.0)
    for _ in range(10):
        run.frame_total()
    ta, tb, tt = [], [], []
    for _ in range(iters):
        t0 = time.perf_counter()
        run.one_call()
        t1 = time.perf_counter()
        run.frame_mp[:] = -1
        msorb._check(run.L.msorb_search_local_points(*run._lp), "msorb_search_local_points")
        t2 = time.perf_counter()
        ta.append(t1 - t0); tb.append(t2 - t1); tt.append(t2 - t0)
    te = []
    for _ in range(iters):
        t0 = time.perf_counter()
        msorb._check(run.L.msorb_extract_stereo_frame(*run._stereo), "msorb_extract_stereo_frame")
        te.append(time.perf_counter() - t0)
    med = lambda v: round(float(np.median(v)) * 1e3, 4)
    print(json.dumps({"keypoints": [int(run.nl.value), int(run.nr.value)], "ms_frame_and_motion_search": med(ta),
                      "ms_search_local_points": med(tb), "ms_frame_total": med(tt), "ms_extract_stereo_frame": med(te),
                      "motion_matches": int(run.nm.value), "local_matches": int(run.nm_lp.value), "iters": iters}))
    run.close()
    ex.close()


if __name__ == "__main__":
    main()
