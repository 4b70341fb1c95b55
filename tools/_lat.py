import sys, os, time, numpy as np
sys.path[:0] = ["ms-slam_amd"]
import msorb
from msorb import synth
cfg = synth.KITTI
ex = msorb.ORBextractor(cfg["nfeatures"], cfg["scale"], cfg["nlevels"], cfg["ini_th"], cfg["min_th"])
imgs = [synth.image(i, cfg["rows"], cfg["cols"]) for i in range(4)]
for i in range(10): ex(imgs[i % 4])
t = []
for i in range(200):
    t0 = time.perf_counter(); ex(imgs[i % 4]); t.append(time.perf_counter() - t0)
print("single-thread msorb_extract median ms:", round(float(np.median(t)) * 1e3, 4), "min", round(min(t) * 1e3, 4))
ex.set_profiling(True)
acc = {}
for i in range(20):
    ex(imgs[i % 4])
    for k, v in ex.stage_ms().items(): acc[k] = acc.get(k, 0) + v / 20
print("stage ms (events):", {k: round(v, 4) for k, v in acc.items()}, "sum", round(sum(acc.values()), 4))
