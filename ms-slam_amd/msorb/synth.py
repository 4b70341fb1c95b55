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


# Input classes of the density sweep (bench.py density_sweep): the default recipe is corner-rich (5.8 % of the
# pixels pass FAST at threshold 20); "low" is the street-imagery regime — 0.6 % corners, a third of the 35 x 35 cells without a
# corner at 20 so that they take the minThFAST retry of ORBextractor.cc:843-847 — and "high" a cluttered one.  (count scale of the three
# shape sizes, amplitude of a shape, noise sigma)
TEXTURE = {"default": (1.0, 70.0, 3.0), "low": (0.2, 50.0, 2.0), "high": (2.5, 95.0, 4.0)}


def _scene(rng, rows, cols, texture="default"):
    density, amp, _ = TEXTURE[texture]
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
        for _ in range(int(round(count * density))):
            cx, cy = rng.uniform(0, cols), rng.uniform(0, rows)
            w, h = rng.uniform(0.4, 1.6, 2) * size
            val = rng.uniform(-amp, amp)
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


def image(seed, rows, cols, sigma=None, texture="default"):
    """One u8 rows x cols image of the given input class (TEXTURE)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    return _finish(rng, _scene(rng, rows, cols, texture), TEXTURE[texture][2] if sigma is None else sigma)


def stereo_pair(seed, rows, cols, max_disp=48.0, sigma=None, texture="default"):
    """(left, right) u8 images; right = left resampled with a smooth positive disparity field."""
    rng = np.random.Generator(np.random.PCG64(seed))
    sigma = TEXTURE[texture][2] if sigma is None else sigma
    scene = _scene(rng, rows, cols, texture)
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


def stereo_batch(n_pairs, rows, cols, seed0=0, texture="default"):
    """uint8 array [2*n_pairs, rows, cols]: L0, R0, L1, R1, ..."""
    out = np.empty((2 * n_pairs, rows, cols), np.uint8)
    for i in range(n_pairs):
        out[2 * i], out[2 * i + 1] = stereo_pair(seed0 + i, rows, cols, texture=texture)
    return out


KITTI_CAM = dict(fx=718.856, fy=718.856, cx=607.1928, cy=185.2157, mbf=386.1448)   # Examples/Stereo/KITTI00-02.yaml


def local_map(seed, kps, desc, depth, scale, m, pose=None, copy_frac=0.7, bad_frac=0.03, spars_frac=0.05, obs_zero_frac=0.15,
              skip_frac=0.1, max_flips=40, cam=KITTI_CAM, bounds=(0.0, 1241.0, 0.0, 376.0)):
    """A tracking-loop workload: m local map points in WORLD coordinates for the camera pose
    (Rcw, tcw) — default identity —, copy_frac of them on the viewing ray of a keypoint (a few pixels off, at its stereo
    depth when it has one) with the keypoint's descriptor up to max_flips bits away, the rest anywhere around the view with
    random descriptors.  Vectorised numpy, seeded.  -> dict(pos_w, normal, max_distance, min_distance, visit, bad,
    sparsified, desc, obs, flags) as msorb_search_local_points / msorb_track_batch take them."""
    rng = np.random.Generator(np.random.PCG64(seed))
    R, t = (np.eye(3, dtype=np.float32), np.zeros(3, np.float32)) if pose is None else pose
    Ow = (-(R.T.astype(np.float64) @ t.astype(np.float64))).astype(np.float32)
    n, nlev = len(kps), len(scale)
    src = rng.integers(0, max(n, 1), m)
    is_copy = (rng.random(m) < copy_frac) & (n > 0)
    kx = kps["x"][src] if n else np.zeros(m, np.float32)
    ky = kps["y"][src] if n else np.zeros(m, np.float32)
    ko = kps["octave"][src] if n else np.zeros(m, np.int32)
    kd = depth[src] if n else np.full(m, -1.0, np.float32)
    u = np.where(is_copy, kx + rng.normal(0, 2.0, m), rng.uniform(bounds[0] - 100, bounds[1] + 100, m))
    v = np.where(is_copy, ky + rng.normal(0, 2.0, m), rng.uniform(bounds[2] - 60, bounds[3] + 60, m))
    z = np.where(is_copy & (kd > 0), kd * rng.uniform(0.97, 1.03, m), rng.uniform(2.0, 70.0, m))
    Pc = np.stack([(u - cam["cx"]) * z / cam["fx"], (v - cam["cy"]) * z / cam["fy"], z], 1)
    Pw = ((Pc - t.astype(np.float64)) @ R.astype(np.float64)).astype(np.float32)
    po = Pw.astype(np.float64) - Ow.astype(np.float64)
    d = np.linalg.norm(po, axis=1) + 1e-9
    nrm = po / d[:, None] + rng.normal(scale=0.35, size=po.shape)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    lvl = np.where(is_copy, ko + rng.integers(0, 2, m), rng.integers(0, nlev, m)).clip(0, nlev - 1)
    maxd = (d * np.asarray(scale, np.float64)[lvl] * rng.uniform(0.93, 1.0, m)).astype(np.float32)
    mind = (maxd / np.float32(scale[-1])).astype(np.float32)
    dsc = rng.integers(0, 256, (m, 32), dtype=np.uint8)
    if n:
        cp = desc[src].copy()
        nf = rng.integers(0, max_flips + 1, m)
        for k in range(max_flips):   # flip k-th random bit of the points that still have flips to spend (repeats may cancel)
            bit = rng.integers(0, 256, m)
            on = (k < nf)
            cp[np.arange(m)[on], (bit >> 3)[on]] ^= (1 << (bit & 7)[on]).astype(np.uint8)
        dsc = np.where(is_copy[:, None], cp, dsc).astype(np.uint8)
    visit = (rng.random(m) >= skip_frac).astype(np.uint8)
    bad = (rng.random(m) < bad_frac).astype(np.uint8)
    spars = (rng.random(m) < spars_frac).astype(np.uint8)
    return dict(pos_w=Pw, normal=nrm.astype(np.float32), max_distance=maxd, min_distance=mind, visit=visit, bad=bad,
                sparsified=spars, desc=dsc, obs=np.where(rng.random(m) < obs_zero_frac, 0, rng.integers(1, 12, m)).astype(np.int32),
                flags=(visit | (bad << 1) | (spars << 2)).astype(np.uint8), Rcw=R, tcw=t, Ow=Ow)


def _quat_xyzw(R):
    """unit quaternion (x, y, z, w) of a rotation matrix (float64 in, as Sophus holds Tcw's rotation)"""
    R = np.asarray(R, np.float64)
    w = np.sqrt(max(0.0, 1.0 + R[0, 0] + R[1, 1] + R[2, 2])) / 2.0
    x = (R[2, 1] - R[1, 2]) / (4.0 * w)
    y = (R[0, 2] - R[2, 0]) / (4.0 * w)
    z = (R[1, 0] - R[0, 1]) / (4.0 * w)
    q = np.array([x, y, z, w])
    return q / np.linalg.norm(q)


def _rot(rx, ry, rz):
    cx, sx, cy, sy, cz, sz = np.cos(rx), np.sin(rx), np.cos(ry), np.sin(ry), np.cos(rz), np.sin(rz)
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


def last_frame(seed, kps, desc, depth, point_frac=0.75, outlier_frac=0.05, obs_zero_frac=0.15, max_flips=30, pixel_sigma=2.5,
               motion=(0.004, 0.003, 0.002, 0.02, 0.01, 0.35), behind_frac=0.01, cam=KITTI_CAM):
    """TrackWithMotionModel's workload: a "last frame" whose keypoints are those of the current frame
    seen from a slightly different pose.  point_frac of the keypoints hold a map point (has_point), placed on the keypoint's
    viewing ray at its stereo depth (or a random depth) in the LAST camera's frame a few pixels off; the current pose Tcw =
    the small motion `motion` (3 rotation angles in rad, 3 translation components in m) applied to the last pose, so the
    points project near — not on — the keypoints.  Descriptors: the keypoint's, up to max_flips bits away.  behind_frac of the
    points sit behind the camera.  -> (dict(has_point, pos_w, octave, angle, desc, obs), q_xyzw, t, forward, backward)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    n = len(kps)
    # last pose: something non-trivial, so that the quaternion action has every term
    Rlw = _rot(0.11, -0.27, 0.05)
    tlw = np.array([0.8, -0.3, 2.1])
    dR = _rot(*motion[:3])
    dt = np.asarray(motion[3:], np.float64)
    Rcw, tcw = dR @ Rlw, dR @ tlw + dt          # Tcw = dT * Tlw
    kx, ky = kps["x"].astype(np.float64), kps["y"].astype(np.float64)
    z = np.where(depth > 0, depth.astype(np.float64) * rng.uniform(0.98, 1.02, n), rng.uniform(3.0, 60.0, n))
    behind = rng.random(n) < behind_frac
    # positions chosen in the CURRENT camera (so the projections land pixel_sigma around the keypoints), then moved to the world
    u = kx + rng.normal(0, pixel_sigma, n)
    v = ky + rng.normal(0, pixel_sigma, n)
    zc = np.where(behind, -z, z)
    Pc = np.stack([(u - cam["cx"]) * zc / cam["fx"], (v - cam["cy"]) * zc / cam["fy"], zc], 1)
    Pw = ((Pc - tcw) @ Rcw).astype(np.float32)
    has = (rng.random(n) < point_frac) & ~(rng.random(n) < outlier_frac)
    cp = desc.copy()
    nf = rng.integers(0, max_flips + 1, n)
    for k in range(max_flips):
        bit = rng.integers(0, 256, n)
        on = k < nf
        cp[np.arange(n)[on], (bit >> 3)[on]] ^= (1 << (bit & 7)[on]).astype(np.uint8)
    # the last frame saw the same corners at (almost) the same orientation and level
    ang = (kps["angle"].astype(np.float64) + rng.normal(0, 4.0, n)) % 360.0
    flip = rng.random(n) < 0.06                  # a few wild orientations: the rotation histogram has something to drop
    ang = np.where(flip, rng.uniform(0, 360, n), ang).astype(np.float32)
    octv = np.clip(kps["octave"].astype(np.int64) + rng.integers(-1, 2, n) * (rng.random(n) < 0.3), 0, 7).astype(np.int32)
    obs = np.where(rng.random(n) < obs_zero_frac, 0, rng.integers(1, 12, n)).astype(np.int32)
    tlc_z = float((Rlw @ (-(Rcw.T @ tcw)) + tlw)[2])   # tlc = Tlw * twc (ORBmatcher.cc:1954-1955)
    mb = cam["mbf"] / cam["fx"]
    # Rcw / tcw / Rlw / tlw ride along for callers that hold the poses as matrices (the host-projected class path)
    return (dict(has_point=has.astype(np.uint8), pos_w=Pw, octave=octv, angle=ang, desc=cp.astype(np.uint8), obs=obs,
                 Rcw=Rcw.astype(np.float32), tcw=tcw.astype(np.float32), Rlw=Rlw.astype(np.float32), tlw=tlw.astype(np.float32)),
            _quat_xyzw(Rcw).astype(np.float32), tcw.astype(np.float32), tlc_z > mb, -tlc_z > mb)
