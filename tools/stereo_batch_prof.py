"""Kernel timing of the batched stereo association (msorb_stereo_matches_batch) on 128 extracted KITTI-like pairs:
cd /tmp && rocprofv3 --kernel-trace --stats -d <out> -- python tools/stereo_batch_prof.py   (GPU box)"""
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
d = torch.from_numpy(np.ascontiguousarray(host)).cuda()
counts, _, k, de = ex.extract_batch(d, (0, 0))
for _ in range(12):
    r = msorb.stereo_matches_batch(ex, counts, k, de, 0.537, 386.1448)
print("ms per batch (last call)", r[3])
