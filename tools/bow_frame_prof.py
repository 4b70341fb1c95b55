"""Kernel timing of the per-frame Frame::ComputeBoW call (msorb_bow_transform, 2000 descriptors, ORBvoc-shaped synthetic vocabulary):
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d <out> -o s -- python tools/bow_frame_prof.py   (GPU box)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "ms-slam_amd"), os.path.join(ROOT, "tests")]
import msorb, bow_cases
voc = bow_cases.make_vocabulary(0, k=10, L=6, stop_frac=0.01)
dev = msorb.Vocabulary(voc["k"], voc["L"], 0, 0, voc["parent"], voc["is_leaf"], voc["descriptors"], voc["weights"])
d = bow_cases.make_features(1, voc, 2000)
for _ in range(5): dev.transform(d)
t = []
for _ in range(50):
    t0 = time.perf_counter(); dev.transform(d); t.append(time.perf_counter() - t0)
print("ms per call (median)", round(float(np.median(t)) * 1e3, 4))
