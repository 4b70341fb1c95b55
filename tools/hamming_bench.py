"""Dense brute-force Hamming top-2 (msorb_hamming_dense_top2_batch) alone: Tpairs/s on the bench shape (128 frames, ~2000 x
~2000 descriptors each) and a check of one frame against the CPU oracle.  usage: python tools/hamming_bench.py [frames] [--popcount]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "ms-slam_amd"), os.path.join(ROOT, "oracle")]
import torch
import msorb
import orb_oracle

popcount = "--popcount" in sys.argv            # the xor / popcount formulation (BASELINE north_star's) instead of the matrix cores
args = [a for a in sys.argv[1:] if not a.startswith("--")]
frames = int(args[0]) if args else 128
form = msorb.DENSE_POPCOUNT if popcount else msorb.DENSE_MATRIX_CORES
rng = np.random.default_rng(0)
cap = 2024
nq = rng.integers(1990, 2016, frames).astype(np.int32)
nt = rng.integers(1990, 2016, frames).astype(np.int32)
q = rng.integers(0, 256, (frames, cap, 32), dtype=np.uint8)
t = rng.integers(0, 256, (frames, cap, 32), dtype=np.uint8)
t[:, :1000] = q[:, :1000] ^ (rng.random((frames, 1000, 32)) < 0.03).astype(np.uint8)   # near copies: small distances, ties
dq, dt = torch.from_numpy(q).cuda(), torch.from_numpy(t).cuda()
dnq, dnt = torch.from_numpy(nq).cuda(), torch.from_numpy(nt).cuda()
msorb.hamming_dense_top2_batch(dq, dt, dnq, dnt, repeats=3, formulation=form)
reps = 30
bi, bd, sd, ms = msorb.hamming_dense_top2_batch(dq, dt, dnq, dnt, repeats=reps, formulation=form)
pairs = int((nq.astype(np.int64) * nt).sum())
wi, wd, ws = orb_oracle.dense_top2(q[3, :nq[3]], t[3, :nt[3]])
ok = (np.array_equal(wi, bi[3, :nq[3]].cpu().numpy()) and np.array_equal(wd, bd[3, :nq[3]].cpu().numpy()) and
      np.array_equal(ws, sd[3, :nq[3]].cpu().numpy()))
# ceiling of the matrix-core kernel: the i8 MFMA rate the chip sustains (4.3 POPS, tools/mfma_rate.hip) / 512 int8 operations per
# pair; --popcount selects the xor / popcount kernel, whose integer-VALU ceiling is 2486 Gpairs/s (2072 with top-2;
# v_xor + v_bcnt cost 7.91 cycles@2.4GHz per instruction pair in a mixed stream, tools/valu_ubench2.hip)
valu = popcount
ceiling = 1024 * 2.4e9 * 64 / (8 * 7.91 + 3 * 4.2) / 1e9 if valu else 4.3e15 / 512 / 1e9
g = pairs * reps / (ms * 1e-3) / 1e9
print(json.dumps({"kernel": "dense_top2_kernel (VALU)" if valu else "dense_top2_mfma_kernel", "gpairs_per_s": round(g, 1),
                  "ms_per_launch": round(ms / reps, 4), "pairs_per_launch": pairs, "matches_oracle": bool(ok),
                  "ceiling_gpairs_per_s": round(ceiling, 1), "frac_of_ceiling": round(g / ceiling, 3)}))
