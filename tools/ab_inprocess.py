#!/usr/bin/env python3
"""In-process A/B of a per-frame variant: two processes of the SAME build differ by up to 6 % in per-frame medians (clocks, the core
the host thread lands on), so a variant worth 1-3 % is only visible when both sides run in ONE process.  The experimental build reads
an environment variable PER CALL (`getenv` in the call path — experiment builds only, the library proper reads its switches once);
this driver flips it between alternating blocks of 100 frames (order reversed every round) and prints the medians per value:
    gpurun -- 'python tools/ab_inprocess.py MSORB_X_SOMETHING 0 1'
Columns: msorb_track_frontend_motion (one call), msorb_extract_stereo_frame, msorb_extract_stereo through the Python mirror."""
import json, os, sys, time
VAR, VALUES = sys.argv[1], sys.argv[2:]
assert len(VALUES) >= 2
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "ms-slam_amd")]
import msorb
from msorb import synth
MBF = 386.1448; MB = MBF / 718.856
cfg = synth.KITTI
L, R = synth.stereo_pair(0, cfg["rows"], cfg["cols"])
ex = msorb.ORBextractor(cfg["nfeatures"], cfg["scale"], cfg["nlevels"], cfg["ini_th"], cfg["min_th"])
kl, dl, kr, dr, ur, dp, oob = ex.extract_stereo(L, R, MB, MBF)
scale = ex.GetScaleFactors(); cam = synth.KITTI_CAM
bounds = (0.0, float(cfg["cols"]), 0.0, float(cfg["rows"]))
last, q, t, fw, bw = synth.last_frame(9500, kl, dl, dp)
mm = msorb.MotionModel.make(q, t, cam["fx"], cam["fy"], cam["cx"], cam["cy"], cam["mbf"], fw, bw)
run = msorb.MotionFrontendRunner(ex, L, R, MB, MBF, mm, last, last["obs"], 7.0)
for _ in range(50): run.one_call()
res = {v: [] for v in VALUES}; res_e = {v: [] for v in VALUES}; res_p = {v: [] for v in VALUES}
for rnd in range(12):
    for v in (VALUES if rnd % 2 == 0 else VALUES[::-1]):
        os.environ[VAR] = v
        for _ in range(5): run.one_call()
        for _ in range(100):
            t0 = time.perf_counter(); run.one_call(); res[v].append(time.perf_counter() - t0)
        for _ in range(100):
            t0 = time.perf_counter(); msorb._check(run.L.msorb_extract_stereo_frame(*run._stereo), "x"); res_e[v].append(time.perf_counter() - t0)
        for _ in range(100):
            t0 = time.perf_counter(); out = ex.extract_stereo(L, R, MB, MBF); res_p[v].append(time.perf_counter() - t0)
for v in VALUES:
    print(VAR, v, "motion call median ms", round(float(np.median(res[v])) * 1e3, 4), "mean", round(float(np.mean(res[v])) * 1e3, 4),
          "| stereo frame median", round(float(np.median(res_e[v])) * 1e3, 4), "| python extract_stereo median", round(float(np.median(res_p[v])) * 1e3, 4), "matches", int(run.nm.value))
