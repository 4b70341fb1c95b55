"""device time of the careful loop's std::sort restatement alone (msorb_debug_std_sort), frame form with / without the in-lane finish"""
import sys, os
import numpy as np
sys.path[:0] = [os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ms-slam_amd")]
import msorb
rng = np.random.default_rng(3)
for n in (12, 30, 48, 64, 65, 100, 150, 200, 300, 434):
    k = (rng.integers(2, 12, n).astype(np.uint32) << 16) | rng.choice(np.array([0, 77, 155, 232, 310, 387, 465], np.uint32), n)
    t = {}
    for lane in (True, False):
        ts = [msorb.debug_std_sort(k, True, lane_sort=lane, timing=True)[2] for _ in range(7)]
        t[lane] = min(ts[2:])
    tb = min(msorb.debug_std_sort(k, False, timing=True)[2] for _ in range(5))
    print(f"n={n:4d}: frame form with lanes {t[True]:6.2f} us, without {t[False]:6.2f} us | batch form {tb:6.2f} us", flush=True)
