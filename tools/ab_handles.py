#!/usr/bin/env python3
"""In-process A/B of per-frame switches that a handle reads ONCE when it is created (MSORB_FRAME_FUSE, MSORB_FRAME_COMPACT, ...): one
extractor handle per setting in ONE process (two processes of the same build differ by up to 6 % in per-frame medians), alternating
blocks of 100 frames, order reversed every round; medians per setting.
    gpurun -- 'python tools/ab_handles.py MSORB_FRAME_FUSE=0,MSORB_FRAME_COMPACT=0 MSORB_FRAME_FUSE=1,MSORB_FRAME_COMPACT=0 MSORB_FRAME_FUSE=1,MSORB_FRAME_COMPACT=1'
Each argument is a comma-separated list of VAR=value (or "default").  Columns: msorb_extract (one image), msorb_extract_stereo (through
the Python mirror), msorb_extract_stereo_frame, msorb_track_frontend_motion (one call).  Results are checked equal across settings."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "ms-slam_amd")]
import msorb
from msorb import synth
SETTINGS = sys.argv[1:]
assert len(SETTINGS) >= 2
MBF = 386.1448; MB = MBF / 718.856
cfg = synth.KITTI
L, R = synth.stereo_pair(0, cfg["rows"], cfg["cols"])
rigs = []
for st in SETTINGS:
    kv = {} if st == "default" else dict(x.split("=") for x in st.split(","))
    old = {k: os.environ.get(k) for k in kv}
    os.environ.update(kv)
    ex = msorb.ORBextractor(cfg["nfeatures"], cfg["scale"], cfg["nlevels"], cfg["ini_th"], cfg["min_th"])
    for k, v in old.items():
        if v is None: os.environ.pop(k, None)
        else: os.environ[k] = v
    kl, dl, kr, dr, ur, dp, oob = ex.extract_stereo(L, R, MB, MBF)
    cam = synth.KITTI_CAM
    last, q, t, fw, bw = synth.last_frame(9500, kl, dl, dp)
    mm = msorb.MotionModel.make(q, t, cam["fx"], cam["fy"], cam["cx"], cam["cy"], cam["mbf"], fw, bw)
    run = msorb.MotionFrontendRunner(ex, L, R, MB, MBF, mm, last, last["obs"], 7.0)
    for _ in range(50): run.one_call()
    rigs.append(dict(ex=ex, run=run, ref=(kl, dl, kr, dr, ur, dp), t={"one": [], "stereo": [], "frame": [], "motion": []}))
for r in rigs[1:]:
    for a, b in zip(r["ref"], rigs[0]["ref"]):
        assert np.array_equal(np.asarray(a).view(np.uint8), np.asarray(b).view(np.uint8)), "the settings disagree on a result"
def timed(fn, n=100):
    out = []
    for _ in range(n):
        t0 = time.perf_counter(); fn(); out.append(time.perf_counter() - t0)
    return out
for rnd in range(12):
    for r in (rigs if rnd % 2 == 0 else rigs[::-1]):
        ex, run = r["ex"], r["run"]
        for _ in range(5): run.one_call()
        r["t"]["motion"] += timed(run.one_call)
        r["t"]["frame"] += timed(lambda: msorb._check(run.L.msorb_extract_stereo_frame(*run._stereo), "x"))
        r["t"]["stereo"] += timed(lambda: ex.extract_stereo(L, R, MB, MBF))
        r["t"]["one"] += timed(lambda: ex(L))
for st, r in zip(SETTINGS, rigs):
    m = {k: round(float(np.median(v)) * 1e3, 4) for k, v in r["t"].items()}
    print(f"{st:60s} one image {m['one']:.4f} | extract_stereo (python) {m['stereo']:.4f} | stereo_frame {m['frame']:.4f} | motion call {m['motion']:.4f} ms  matches {int(r['run'].nm.value)}", flush=True)
