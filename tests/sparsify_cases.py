"""Synthetic MapSparsification windows: 30 keyframes x ~1000 tracked points, 64x48 grid,
observation counts ~Geometric(mean 8), ~100 keyframes outside the window."""
import numpy as np


def window(seed, n_window=30, n_outside=100, n_points=6000, slots_per_kf=2000, tracked_frac=0.5, grid=(64, 48)):
    rng = np.random.Generator(np.random.PCG64(seed))
    n_kf = n_window + n_outside
    window_ids = np.sort(rng.choice(n_kf, n_window, replace=False))
    in_window = np.zeros(n_kf, np.uint8)
    in_window[window_ids] = 1
    bad = rng.random(n_points) < 0.05
    obs_lists = [set() for _ in range(n_points)]
    kf_slot_begin, slot_point, slot_cell = [0], [], []
    for k in window_ids:
        cells = np.sort(rng.integers(0, grid[0] * grid[1], slots_per_kf))   # grid walk order: col-major cell id
        tracked = rng.random(slots_per_kf) < tracked_frac
        pts = rng.choice(n_points, slots_per_kf, replace=False)
        for c, t, p in zip(cells, tracked, pts):
            if t:
                obs_lists[p].add(int(k))
                slot_point.append(-1 if bad[p] else int(p))
            else:
                slot_point.append(-1)
            slot_cell.append(int(c))
        kf_slot_begin.append(len(slot_point))
    outside = np.nonzero(in_window == 0)[0]
    for p in range(n_points):
        extra = min(rng.geometric(1 / 8.0) - 1, len(outside))
        if extra > 0:
            obs_lists[p].update(int(x) for x in rng.choice(outside, extra, replace=False))
    obs_begin, obs_kf = [0], []
    for p in range(n_points):
        obs_kf.extend(sorted(obs_lists[p]))
        obs_begin.append(len(obs_kf))
    point_nobs = np.diff(obs_begin).astype(np.int32)
    kf_num_mps = rng.integers(200, 1500, n_kf).astype(np.int32)
    return dict(kf_slot_begin=np.array(kf_slot_begin, np.int32), slot_point=np.array(slot_point, np.int32),
                slot_cell=np.array(slot_cell, np.int32), point_nobs=point_nobs, obs_begin=np.array(obs_begin, np.int32),
                obs_kf=np.array(obs_kf, np.int32), kf_in_window=in_window, kf_num_mps=kf_num_mps)
