"""Synthetic tracking-loop workloads shared by the CPU and GPU matcher tests."""
import numpy as np

from msorb import synth

KITTI_FX, KITTI_BF = 718.856, 386.1448   # Examples/Stereo/KITTI00-02.yaml (fx, bf)


def flip_bits(rng, desc, max_flips):
    out = desc.copy()
    for i in range(len(out)):
        nf = rng.integers(0, max_flips + 1)
        bits = rng.choice(256, nf, replace=False)
        for b in bits:
            out[i, b >> 3] ^= 1 << (b & 7)
    return out


def map_point_table(rng, kps, desc, ur, scale, M, obs_zero_frac=0.15, sparsified_frac=0.05, dup=3):
    """M map points: half are (noisy) copies of frame descriptors projected near their keypoint — several map
    points per keypoint so that claims collide — half are random."""
    n = len(kps)
    src = rng.integers(0, n, M)
    is_copy = rng.random(M) < 0.6
    d = np.where(is_copy[:, None], flip_bits(rng, desc[src], 40), rng.integers(0, 256, (M, 32), dtype=np.uint8))
    d = d.astype(np.uint8)
    px = kps["x"][src] + rng.normal(0, 3, M).astype(np.float32)
    py = kps["y"][src] + rng.normal(0, 3, M).astype(np.float32)
    level = np.clip(kps["octave"][src] + rng.integers(0, 2, M), 0, len(scale) - 1).astype(np.int32)
    pxr = np.where(ur[src] > 0, ur[src] + rng.normal(0, 2, M), px - 20).astype(np.float32)
    return dict(
        track_in_view=(rng.random(M) < 0.9).astype(np.uint8), bad=(rng.random(M) < 0.03).astype(np.uint8),
        sparsified=(rng.random(M) < sparsified_frac).astype(np.uint8), proj_x=px.astype(np.float32),
        proj_y=py.astype(np.float32), proj_xr=pxr, track_depth=rng.uniform(2, 80, M).astype(np.float32), level=level,
        view_cos=rng.uniform(0.99, 1.0, M).astype(np.float32), desc=d,
        obs=np.where(rng.random(M) < obs_zero_frac, 0, rng.integers(1, 12, M)).astype(np.int32))


def last_frame_table(rng, kps, desc, ur, scale, NL, obs_zero_frac=0.2):
    n = len(kps)
    src = rng.integers(0, n, NL)
    d = flip_bits(rng, desc[src], 30)
    u = kps["x"][src] + rng.normal(0, 4, NL)
    v = kps["y"][src] + rng.normal(0, 4, NL)
    return dict(valid=(rng.random(NL) < 0.85).astype(np.uint8), u=u.astype(np.float32), v=v.astype(np.float32),
                ur=np.where(ur[src] > 0, ur[src] + rng.normal(0, 2, NL), u - 15).astype(np.float32),
                octave=kps["octave"][src].astype(np.int32), angle=(kps["angle"][src] + rng.normal(0, 8, NL)).astype(np.float32) % 360,
                desc=d, mp=np.arange(NL, dtype=np.int32),
                obs=np.where(rng.random(NL) < obs_zero_frac, 0, rng.integers(1, 9, NL)).astype(np.int32))
