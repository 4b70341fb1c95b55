"""One-off fuzz: random geometries / pitches / alignments through the batch kernels vs the oracle."""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "ms-slam_amd"), os.path.join(ROOT, "oracle")]
import torch, msorb, orb_oracle
from msorb import synth
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
bad = 0
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 20):
    rows = int(rng.integers(240, 520)); cols = int(rng.integers(330, 1400))
    nfeat = int(rng.choice([500, 1000, 2000])); nlev = int(rng.choice([4, 6, 8]))
    pad = int(rng.choice([0, 4, 8, 16, 32, 64])); pad += (-(cols + pad)) % int(rng.choice([4, 16, 64]))
    off = int(rng.choice([0, 0, 4, 16, 32]))
    n = int(rng.choice([16, 17, 32, 64]))
    pitch = cols + pad
    ex = msorb.ORBextractor(nfeat, 1.2, nlev, 20, 7)
    ref = orb_oracle.OracleExtractor(nfeat, 1.2, nlev, 20, 7)
    try:
        imgs = np.stack([synth.image(5000 + it * 7 + (i % 3), rows, cols) for i in range(n)])
        flat = torch.zeros(n * rows * pitch + 64, dtype=torch.uint8, device="cuda")
        store = flat[off:off + n * rows * pitch].view(n, rows, pitch)
        view = store[:, :, :cols]
        view.copy_(torch.from_numpy(imgs).cuda())
        counts, mono, d_kps, d_desc = ex.extract_batch(view)
        kps = msorb.keypoints_from_device(d_kps, counts)
        desc = d_desc.cpu().numpy()
        ok = True
        for i in (0, 1, 2, n - 1):
            rmono, rkps, rdesc = ref(imgs[i])
            same = counts[i] == len(rkps) and mono[i] == rmono and np.array_equal(kps[i].view(np.uint8), rkps.view(np.uint8)) and np.array_equal(desc[i, :counts[i]], rdesc)
            if i == n - 1:
                for l in range(1, nlev):
                    same = same and np.array_equal(ex.debug_level(i, l), ref.level(l)) and np.array_equal(ex.debug_level(i, l, blurred=True), ref.level(l, blurred=True))
            ok = ok and bool(same)
        print(it, rows, cols, "pitch", pitch, "off", off, "n", n, "nfeat", nfeat, "levels", nlev, "OK" if ok else "MISMATCH", flush=True)
        bad += 0 if ok else 1
    except Exception as e:
        print(it, rows, cols, pitch, off, n, nfeat, nlev, "EXC", repr(e)[:200], flush=True)
        bad += 1
    finally:
        ex.close()
print("bad", bad)
