"""Seeded synthetic stand-ins for the datasets BASELINE.json names (no datasets exist on the
build or GPU boxes).  SURVEY.md §8d: sum of random rectangles/discs at three scales + low-frequency
shading + Gaussian pixel noise, clipped to u8, so FAST fires at both thresholds (20 and 7); the
right eye is the left shifted by a smooth disparity field so stereo matching has work to do.

Pure numpy with a PCG64 Generator: the same seed gives the same bytes on every machine.
"""
import numpy as np

KITTI = dict(rows=376, cols=1241, nfeatures=2000, scale=1.2, nlevels=8, ini_th=20, min_th=7)
EUROC = dict(rows=480, cols=752, nfeatures=1000, scale=1.2, nlevels=8, ini_th=20, min_th=7)
EUROC_YAML = dict(EUROC, nfeatures=1200)
FOURSEASONS = dict(rows=400, cols=800, nfeatures=2000, scale=1.2, nlevels=8, ini_th=20, min_th=7)


def _scene(rng, rows, cols):
    img = np.full((rows, cols), 110.0, np.float32)
    yy, xx = np.mgrid[0:rows, 0:cols].astype(np.float32)
    # low-frequency shading
    for _ in range(4):
        fx, fy = rng.uniform(0.002, 0.02, 2)
        img += rng.uniform(5, 18) * np.sin(xx * fx * 6.28 + rng.uniform(0, 6.28)) * np.cos(
            yy * fy * 6.28 + rng.uniform(0, 6.28))
    # shapes at three scales
    area = rows * cols
    for size, count in ((60, area // 9000), (22, area // 1800), (8, area // 500)):
        for _ in range(int(count)):
            cx, cy = rng.uniform(0, cols), rng.uniform(0, rows)
            w, h = rng.uniform(0.4, 1.6, 2) * size
            val = rng.uniform(-70, 70)
            x0, x1 = int(max(0, cx - w)), int(min(cols, cx + w))
            y0, y1 = int(max(0, cy - h)), int(min(rows, cy + h))
            if x1 <= x0 or y1 <= y0:
                continue
            if rng.uniform() < 0.5:
                img[y0:y1, x0:x1] += val
            else:
                sub = ((xx[y0:y1, x0:x1] - cx) / max(w, 1)) ** 2 + ((yy[y0:y1, x0:x1] - cy) / max(h, 1)) ** 2 < 1
                img[y0:y1, x0:x1] += val * sub
    return img


def _finish(rng, img, sigma=3.0):
    img = img + rng.normal(0, sigma, img.shape).astype(np.float32)
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def image(seed, rows, cols, sigma=3.0):
    """One u8 rows x cols image."""
    rng = np.random.Generator(np.random.PCG64(seed))
    return _finish(rng, _scene(rng, rows, cols), sigma)


def stereo_pair(seed, rows, cols, max_disp=48.0, sigma=3.0):
    """(left, right) u8 images; right = left resampled with a smooth positive disparity field."""
    rng = np.random.Generator(np.random.PCG64(seed))
    scene = _scene(rng, rows, cols)
    yy, xx = np.mgrid[0:rows, 0:cols].astype(np.float32)
    disp = max_disp * (0.15 + 0.85 * (yy / rows)) * (0.8 + 0.2 * np.sin(xx / cols * 3.1))
    # right(x) = left(x + d): a point at uL in the left eye appears at uR = uL - d
    xs = np.clip(xx + disp, 0, cols - 1)
    x0 = np.floor(xs).astype(np.int64)
    x1 = np.minimum(x0 + 1, cols - 1)
    f = xs - x0
    rows_idx = yy.astype(np.int64)
    right = scene[rows_idx, x0] * (1 - f) + scene[rows_idx, x1] * f
    left = _finish(rng, scene, sigma)
    right = _finish(rng, right.astype(np.float32), sigma)
    return left, right


def stereo_batch(n_pairs, rows, cols, seed0=0):
    """uint8 array [2*n_pairs, rows, cols]: L0, R0, L1, R1, ..."""
    out = np.empty((2 * n_pairs, rows, cols), np.uint8)
    for i in range(n_pairs):
        out[2 * i], out[2 * i + 1] = stereo_pair(seed0 + i, rows, cols)
    return out
