"""One-off fuzz: random ORB parameters (features, scale factor, levels, thresholds) and image statistics, per-frame and batch
paths vs the oracle."""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "ms-slam_amd"), os.path.join(ROOT, "oracle")]
import torch, msorb, orb_oracle
from msorb import synth
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
bad = 0
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 20):
    rows = int(rng.integers(200, 500)); cols = int(rng.integers(300, 1300))
    nfeat = int(rng.choice([50, 300, 777, 1000, 1500, 2000, 3000, 5000, 10000]))
    sf = float(rng.choice([1.1, 1.2, 1.25, 1.3, 1.5, 2.0])); nlev = int(rng.integers(1, 9))
    ini = int(rng.integers(5, 60)); mn = int(rng.integers(1, ini + 1))
    kind = int(rng.integers(0, 4))
    n = int(rng.choice([16, 24]))
    try:
        ex = msorb.ORBextractor(nfeat, sf, nlev, ini, mn)
    except Exception as e:
        print(it, "ctor refused", nfeat, sf, nlev, ini, mn, repr(e)[:120]); continue
    ref = orb_oracle.OracleExtractor(nfeat, sf, nlev, ini, mn)
    try:
        base = [synth.image(7000 + it * 5 + i, rows, cols) for i in range(3)]
        def shape(a, k):
            if k == 1: return (a.astype(np.int32) // 6 + 100).astype(np.uint8)            # low contrast: threshold fallback
            if k == 2: return rng.integers(0, 256, a.shape, dtype=np.uint8)               # noise: saturated cells
            if k == 3: b = a.copy(); b[:, : cols // 2] = 77; return b                      # half flat
            return a
        imgs = np.stack([shape(base[i % 3], kind if i % 2 == 0 else 0) for i in range(n)])
        d = torch.from_numpy(imgs).cuda()
        try:
            counts, mono, d_kps, d_desc = ex.extract_batch(d)
        except msorb.MsorbError as e:
            print(it, "batch refused", rows, cols, nfeat, sf, nlev, ini, mn, kind, repr(e)[:160]); ex.close(); continue
        kps = msorb.keypoints_from_device(d_kps, counts)
        desc = d_desc.cpu().numpy()
        ok = True
        for i in (0, 1, 2, n - 1):
            rmono, rkps, rdesc = ref(imgs[i])
            same = counts[i] == len(rkps) and mono[i] == rmono and np.array_equal(kps[i].view(np.uint8), rkps.view(np.uint8)) and np.array_equal(desc[i, :counts[i]], rdesc)
            ok = ok and bool(same)
            if i == 0:
                _, k1, d1 = ex(imgs[i])                                                   # per-frame path
                ok = ok and np.array_equal(k1.view(np.uint8), rkps.view(np.uint8)) and np.array_equal(d1, rdesc)
        print(it, rows, cols, "nfeat", nfeat, "sf", sf, "levels", nlev, "th", ini, mn, "kind", kind, "kp", int(counts[0]), "OK" if ok else "MISMATCH", flush=True)
        bad += 0 if ok else 1
    except Exception as e:
        print(it, rows, cols, nfeat, sf, nlev, ini, mn, kind, "EXC", repr(e)[:300], flush=True)
        bad += 1
    finally:
        ex.close()
print("bad", bad)
