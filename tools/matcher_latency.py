#!/usr/bin/env python3
"""Wall time per call of the matcher entries that bench.py does not time, through the Python mirror (its ~15-25 us of ctypes / numpy
overhead included), with the CPU oracle on the same inputs beside it (one host core): the LocalMapping / LoopClosing / relocalisation /
initialisation searches.  A GPU entry that is not clearly ahead of the scalar CPU here is latency-bound in its host part — the reason this
tool exists (SearchForInitialization read 1.36 ms against the oracle's 0.68 before it took the ranked lists).
    gpurun -- 'python tools/matcher_latency.py' """
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "ms-slam_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import msorb, orb_oracle as oracle
from msorb import synth
import matcher_cases as mc
import bow_cases
import test_matcher_a17_gpu as ta

def med(fn, reps):
    for _ in range(3): fn()
    t = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); t.append(time.perf_counter() - t0)
    return float(np.median(t)) * 1e3

def row(name, g, c, greps=40, creps=5):
    a, b = med(g, greps), med(c, creps)
    print(f"{name:58s} gpu {a:8.3f} ms   cpu oracle {b:8.3f} ms   x{b / a:6.1f}", flush=True)

cfg = synth.KITTI
A = synth.image(21, cfg["rows"], cfg["cols"]); B = np.clip(np.roll(A, (2, 5), (0, 1)).astype(np.int32) + np.random.default_rng(5).integers(-3, 4, A.shape), 0, 255).astype(np.uint8)
ex = msorb.ORBextractor(2000, 1.2, 8, 20, 7)
_, k1, d1 = ex(A); _, k2, d2 = ex(B); scale = np.asarray(ex.GetScaleFactors(), np.float32); ex.close()
s = dict(cfg=cfg, k1=k1, d1=d1, k2=k2, d2=d2, scale=scale)
bounds = (0.0, float(cfg["cols"]), 0.0, float(cfg["rows"]))
rng = np.random.Generator(np.random.PCG64(1))
f1, r1 = ta._kf(msorb, oracle, s, 1); f2, r2 = ta._kf(msorb, oracle, s, 2)
p1 = ta._points_from(rng, k1, d1, (5, 2), 1.5, 20); p2 = ta._points_from(rng, k2, d2, (-5, -2), 1.5, 20)
row("SearchBySim3 (2 x 2000 points, th 7.5)", lambda: msorb.search_by_sim3(f1, f2, p1, p2, 7.5), lambda: oracle.search_by_sim3(r1, r2, p1, p2, 7.5))
src = rng.integers(0, len(k1), 5000); pts = ta._points_from(rng, k1[src], d1[src], (5, 2), 1.5, 30)
row("Fuse(pKF, Scw, ...) search (5000 points, th 4)", lambda: msorb.fuse_sim3_search(f2, pts, 4.0), lambda: oracle.fuse_sim3_search(r2, pts, 4.0))
inv = (np.float32(1) / (scale * scale)).astype(np.float32)
ur = (pts["u"] - 20).astype(np.float32); rad = (np.float32(3.0) * scale[pts["level"]]).astype(np.float32)
row("Fuse(pKF, vpMapPoints, th) search (5000 points, th 3)", lambda: f2.FuseSearch(inv, pts["valid"], pts["u"], pts["v"], ur, pts["level"], rad, pts["desc"]),
    lambda: r2.FuseSearch(inv, pts["valid"], pts["u"], pts["v"], ur, pts["level"], rad, pts["desc"]))
src4 = rng.integers(0, len(k1), 4000); pl = ta._points_from(rng, k1[src4], d1[src4], (5, 2), 1.5, 35); ok = (rng.random(len(k2)) < 0.6).astype(np.uint8)
row("SearchByProjection loop form (4000 points, th 8)", lambda: msorb.search_by_projection_loop(f2, pl, ok, 8.0, 75.0), lambda: oracle.search_by_projection_loop(r2, pl, ok, 8.0, 75.0))
N = len(k2)
t = mc.last_frame_table(rng, k2, d2, np.full(N, -1, np.float32), scale, 3000)
ps = dict(valid=t["valid"], u=t["u"], v=t["v"], level=np.clip(t["octave"] + rng.integers(0, 2, 3000), 0, 7).astype(np.int32), desc=t["desc"], mp=t["mp"])
row("SearchByProjection Sim3 form (3000 points, th 8)", lambda: f2.SearchByProjection_sim3(ps, np.full(N, -1, np.int32), 8.0, 75.0), lambda: r2.SearchByProjection_sim3(ps, np.full(N, -1, np.int32), 8.0, 75.0))
pk = dict(valid=t["valid"], u=t["u"], v=t["v"], level=t["octave"], angle=t["angle"], desc=t["desc"], mp=t["mp"])
row("SearchByProjection(F, pKF, sFound) relocalisation (3000, th 10)", lambda: f2.SearchByProjection_kf(pk, np.full(N, -1, np.int32), 10.0, 100, True),
    lambda: r2.SearchByProjection_kf(pk, np.full(N, -1, np.int32), 10.0, 100, True))
prev = np.stack([k1["x"], k1["y"]], 1).astype(np.float32)
row("SearchForInitialization (2000 features, window 100)", lambda: msorb.search_for_initialization(f1, f2, prev.copy(), 100, 0.9, True),
    lambda: oracle.search_for_initialization(r1, r2, prev.copy(), 100, 0.9, True))
sizes = np.random.default_rng(0).integers(2, 12, 2000).tolist()
desc, ob = bow_cases.make_observations(7, sizes)
row("ComputeDistinctiveDescriptors (2000 points, 2-11 observations)", lambda: msorb.distinctive_descriptors(desc, ob), lambda: oracle.distinctive_descriptors(desc, ob))
f1.close(); f2.close()

# ---- the two-camera arms (tests/test_matcher_rig_gpu.py's synthetic rigs) ----
import test_matcher_rig_gpu as tr
import bow_match_cases as bmc
R = tr.make_rig(oracle, 1, 1500, 1400, 4096)
ofl, ofr = oracle.OracleFrame(R["kl"], R["dl"], None, tr.BOUNDS, tr.SCALE), oracle.OracleFrame(R["kr"], R["dr"], None, tr.BOUNDS, tr.SCALE)
dfl, dfr = msorb.Frame(R["kl"], R["dl"], None, tr.BOUNDS, tr.SCALE), msorb.Frame(R["kr"], R["dr"], None, tr.BOUNDS, tr.SCALE)
row("two cameras: SearchByProjection(F, 4096 map points)", lambda: msorb.search_by_projection_mps_rig(dfl, dfr, R["mp"], R["l2r"], R["r2l"], R["frame_mp"].copy(), 1.0, False, 40.0, 0.8),
    lambda: oracle.search_by_projection_mps_rig(ofl, ofr, R["mp"], R["l2r"], R["r2l"], R["frame_mp"].copy(), 1.0, False, 40.0, 0.8))
R2, last, cur = tr.make_last_table(oracle, 51, 1500, 1400, 1800, 7.0)
ofl2, ofr2 = oracle.OracleFrame(R2["kl"], R2["dl"], None, tr.BOUNDS, tr.SCALE), oracle.OracleFrame(R2["kr"], R2["dr"], None, tr.BOUNDS, tr.SCALE)
dfl2, dfr2 = msorb.Frame(R2["kl"], R2["dl"], None, tr.BOUNDS, tr.SCALE), msorb.Frame(R2["kr"], R2["dr"], None, tr.BOUNDS, tr.SCALE)
row("two cameras: SearchByProjection(Current, Last) (1800 points)", lambda: msorb.search_by_projection_frames_rig(dfl2, dfr2, last, cur.copy(), 7.0, False, False, True),
    lambda: oracle.search_by_projection_frames_rig(ofl2, ofr2, last, cur.copy(), 7.0, False, False, True))
pb = tr._bow_rig_pair(1, 1200, 2400, 1300)
row("two cameras: SearchByBoW(pKF, F) (1200 x 2400)", lambda: msorb.search_by_bow_rig(pb, 1300, 50, 0.7, True), lambda: oracle.search_by_bow_rig(pb, 1300, 50, 0.7, True))
pt = bmc.make_pair(301, 1200, 1300, n_nodes=25, flip=30, dup_frac=0.25)
acc = tr._accept_fn(1, 1200, 1300, 0.5)
row("two cameras: SearchForTriangulation with a Python callback", lambda: msorb.search_for_triangulation_cb(pt, acc, 50, True), lambda: oracle.search_for_triangulation_rig(pt, acc, False, True))
for f in (dfl, dfr, dfl2, dfr2): f.close()
