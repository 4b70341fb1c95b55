#!/usr/bin/env python3
"""In-process A/B of a per-call switch (an MSORB_AB_BUILD library: `make -C ms-slam_amd/csrc EXTRA=-DMSORB_AB_BUILD`) on the three
input classes of msorb/synth.py: msorb_extract_stereo on one handle, alternating blocks of 100 frames per value, order reversed every
round; medians per (class, value).     gpurun -- 'python tools/ab_classes.py MSORB_QT_LANE_SORT 0 1'"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "ms-slam_amd")]
import msorb
from msorb import synth
VAR, VALUES = sys.argv[1], sys.argv[2:]
MBF = 386.1448; MB = MBF / 718.856
cfg = synth.KITTI
ex = msorb.ORBextractor(cfg["nfeatures"], cfg["scale"], cfg["nlevels"], cfg["ini_th"], cfg["min_th"])
for tex in ("low", "default", "high"):
    L, R = synth.stereo_pair(1000, cfg["rows"], cfg["cols"], texture=tex)
    res = {v: [] for v in VALUES}
    one = {v: [] for v in VALUES}
    ref = None
    for rnd in range(10):
        for v in (VALUES if rnd % 2 == 0 else VALUES[::-1]):
            os.environ[VAR] = v
            for _ in range(5): out = ex.extract_stereo(L, R, MB, MBF)
            sig = [np.asarray(a).tobytes() for a in out[:6]]
            assert ref is None or sig == ref, "the values disagree on a result"
            ref = sig
            for _ in range(100):
                t0 = time.perf_counter(); ex.extract_stereo(L, R, MB, MBF); res[v].append(time.perf_counter() - t0)
            for _ in range(100):
                t0 = time.perf_counter(); ex(L); one[v].append(time.perf_counter() - t0)
    for v in VALUES:
        print(f"{tex:8s} {VAR}={v}: stereo frame median {np.median(res[v]) * 1e3:.4f} ms (mean {np.mean(res[v]) * 1e3:.4f}) | one image {np.median(one[v]) * 1e3:.4f} ms | keypoints {len(out[0]) + len(out[2])}", flush=True)
