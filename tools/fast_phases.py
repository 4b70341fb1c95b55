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
for _ in range(2):
    ex.extract_batch(img, (0, 0))
acc = 0
for _ in range(5):
    ex.extract_batch(img, (0, 0))
    acc += ex.stage_ms()["fast"]
print("stop", os.environ.get("MSORB_FAST_DEBUG_STOP", "0"), "fast_ms", round(acc / 5, 4))
