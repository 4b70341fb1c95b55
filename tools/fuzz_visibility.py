"""One-off fuzz: msorb_visibility_csr against the oracle on random windows (sizes, tracked fractions, maps with more keyframes than the\nLDS forms hold): python tools/fuzz_visibility.py <seed> <cases> on the GPU box."""
import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "ms-slam_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import msorb, orb_oracle
import sparsify_cases as sc
KEYS = None
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
bad = 0
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 40):
    kw = dict(n_window=int(rng.integers(1, 40)), n_outside=int(rng.choice([0, 1, 7, 100, 400, 13000])), n_points=int(rng.choice([1, 50, 1000, 8000, 30000])),
              slots_per_kf=int(rng.choice([1, 40, 257, 1024, 2000, 3000])), tracked_frac=float(rng.choice([0.0, 0.1, 0.5, 0.9, 1.0])))
    kw["slots_per_kf"] = min(kw["slots_per_kf"], kw["n_points"])
    w = sc.window(1000 + it, **kw)
    got = msorb.visibility_csr(N=100, **w); want = orb_oracle.visibility_csr(N=100, **w)
    ok = (got["n_cols"], got["n_rows"], got["n_max_obs"]) == (want["n_cols"], want["n_rows"], want["n_max_obs"]) and all(np.array_equal(got[k], want[k]) for k in want if isinstance(want[k], np.ndarray))
    if not ok: bad += 1; print("MISMATCH", kw)
print("cases", it + 1, "bad", bad)
