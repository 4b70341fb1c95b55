"""Pyramid-only loop for kernel traces: 256 KITTI images, msorb_pyramid_batch x N."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "ms-slam_amd")]
import torch
import msorb
from msorb import synth
cfg = synth.KITTI
ex = msorb.ORBextractor(2000, 1.2, 8, 20, 7)
base = synth.stereo_batch(8, cfg["rows"], cfg["cols"], seed0=0)
host = np.concatenate([base] * 16)
pitch = (cfg["cols"] + 63) // 64 * 64
st = torch.zeros((256, cfg["rows"], pitch), dtype=torch.uint8, device="cuda")
img = st[:, :, :cfg["cols"]]
img.copy_(torch.from_numpy(host).cuda())
ex.set_overlap(1, False)
ex.set_profiling(True)
acc = 0
for i in range(12):
    ex.extract_batch(img, (0, 0))
    if i >= 2:
        acc += ex.stage_ms()["pyramid"]
print("pyramid_ms", round(acc / 10, 4))
