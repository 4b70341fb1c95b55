#!/usr/bin/env python3
"""Per-frame (latency) view of the path — BASELINE.json configs[2] and configs[4] — next to bench.py's batched
throughput: host image in, host features out, the way Tracking.cc drives it (one stereo frame at a time).

  frame = 2 x msorb_extract (left/right handles on two host threads, Frame.cc:122-125)
        + msorb_stereo_matches (Frame::ComputeStereoMatches)
        + msorb_search_by_projection_mps with 4096 synthetic map points (Tracking::SearchLocalPoints)
  window = msorb_visibility_csr on a 30-keyframe sparsification window (configs[4])

Prints one JSON object; PCIe copies and host replay are included (these are not the `value` of bench.py)."""
import json, os, sys, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "ms-slam_amd"), os.path.join(ROOT, "tests")]
import msorb
from msorb import synth
import matcher_cases as mc
import sparsify_cases as sc


def main(frames=30):
    cfg = synth.KITTI
    exl = msorb.ORBextractor(cfg["nfeatures"], cfg["scale"], cfg["nlevels"], cfg["ini_th"], cfg["min_th"])
    exr = msorb.ORBextractor(cfg["nfeatures"], cfg["scale"], cfg["nlevels"], cfg["ini_th"], cfg["min_th"])
    pairs = [synth.stereo_pair(900 + i, cfg["rows"], cfg["cols"]) for i in range(4)]
    mbf, mb = mc.KITTI_BF, mc.KITTI_BF / mc.KITTI_FX
    rng = np.random.Generator(np.random.PCG64(0))
    t_ext, t_st, t_sp, t_fs = [], [], [], []
    res = {}
    for f in range(frames + 3):
        L, R = pairs[f % len(pairs)]
        out = [None, None]
        t0 = time.perf_counter()
        th = [threading.Thread(target=lambda e=e, im=im, k=k: out.__setitem__(k, e(im))) for k, (e, im) in enumerate(((exl, L), (exr, R)))]
        [t.start() for t in th]; [t.join() for t in th]
        t1 = time.perf_counter()
        (_, kl, dl), (_, kr, dr) = out
        ur, dp, _ = msorb.stereo_matches(exl, exr, kl, dl, kr, dr, mb, mbf)
        t2 = time.perf_counter()
        frame = msorb.Frame(kl, dl, ur, (0.0, float(cfg["cols"]), 0.0, float(cfg["rows"])), exl.GetScaleFactors())
        t3 = time.perf_counter()
        mp = mc.map_point_table(rng, kl, dl, ur, exl.GetScaleFactors(), 4096)
        fm = np.full(len(kl), -1, np.int32)
        t4 = time.perf_counter()
        n = frame.SearchByProjection_mps(mp, fm, 3.0)
        t5 = time.perf_counter()
        frame.close()
        if f >= 3:
            t_ext.append(t1 - t0); t_st.append(t2 - t1); t_fs.append(t3 - t2); t_sp.append(t5 - t4)
        res.update(keypoints_per_eye=int(len(kl)), stereo_matches=int((ur > 0).sum()), projection_matches=int(n))
    w = sc.window(3)
    msorb.visibility_csr(N=100, **w)
    tv = []
    for _ in range(10):
        t0 = time.perf_counter(); m = msorb.visibility_csr(N=100, **w); tv.append(time.perf_counter() - t0)
    med = lambda v: round(float(np.median(v)) * 1e3, 3)
    res.update(ms_extract_stereo_pair=med(t_ext), ms_stereo_matches=med(t_st), ms_frame_grid_upload=med(t_fs),
               ms_search_by_projection_4096=med(t_sp), ms_visibility_csr_window30=med(tv),
               visibility_rows=int(m["n_rows"]), visibility_cols=int(m["n_cols"]), frames=frames,
               note="host-inclusive wall times per stereo frame (median); not bench.py's batched value")
    print(json.dumps(res))

if __name__ == "__main__":
    main()
