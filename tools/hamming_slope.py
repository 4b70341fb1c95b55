import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "ms-slam_amd")]
import torch, msorb
rng = np.random.default_rng(0)
F, N = 128, 2048
dq = torch.from_numpy(rng.integers(0, 256, (F, N, 32), dtype=np.uint8)).cuda()
dt = torch.from_numpy(rng.integers(0, 256, (F, N, 32), dtype=np.uint8)).cuda()
for nq, nt in ((2048, 2048), (2048, 1024), (2048, 512), (2048, 128), (2048, 32), (2000, 2000)):
    cq = torch.full((F,), nq, dtype=torch.int32, device="cuda"); ct = torch.full((F,), nt, dtype=torch.int32, device="cuda")
    msorb.hamming_dense_top2_batch(dq, dt, cq, ct, repeats=2)
    _, _, _, ms = msorb.hamming_dense_top2_batch(dq, dt, cq, ct, repeats=20)
    print(nq, nt, "ms/launch %.4f" % (ms / 20), "Tpairs/s %.2f" % (F * nq * nt / (ms / 20 * 1e-3) / 1e12))
