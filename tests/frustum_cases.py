"""Seeded camera poses + local-map point clouds for Frame::isInFrustum tests (KITTI-like pinhole)."""
import numpy as np

KITTI_CAM = dict(fx=718.856, fy=718.856, cx=607.1928, cy=185.2157, bounds=(0.0, 1241.0, 0.0, 376.0), mbf=386.1448)


def pose(seed):
    """Random rotation (float32 3x3), translation; Ow = -Rcw^T tcw computed in float32 like Sophus would hand it over."""
    rng = np.random.default_rng(seed)
    q = rng.normal(size=4)
    q /= np.linalg.norm(q)
    w, x, y, z = q
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                  [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                  [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]]).astype(np.float32)
    t = rng.normal(scale=5.0, size=3).astype(np.float32)
    Ow = (-(R.T.astype(np.float64) @ t.astype(np.float64))).astype(np.float32)
    return R, t, Ow


def points(seed, n, R, t, Ow):
    """World points: 70 % placed in front of the camera inside / around the image, the rest anywhere; normals roughly
    towards the camera for most; distance ranges around the true distance for most; a block of hard edge cases."""
    rng = np.random.default_rng(seed)
    c = dict(KITTI_CAM)
    z = rng.uniform(0.5, 80.0, n)
    u = rng.uniform(-200, 1441, n)
    v = rng.uniform(-100, 476, n)
    Pc = np.stack([(u - c["cx"]) * z / c["fx"], (v - c["cy"]) * z / c["fy"], z], 1)
    anywhere = rng.random(n) < 0.3
    Pc[anywhere] = rng.normal(scale=30.0, size=(int(anywhere.sum()), 3))
    Pw = (Pc - t.astype(np.float64)) @ R.astype(np.float64)          # R^T (Pc - t)
    Pw = Pw.astype(np.float32)
    to_cam = Ow.astype(np.float64) - Pw
    d = np.linalg.norm(to_cam, axis=1) + 1e-9
    nrm = -(to_cam / d[:, None]) * -1.0                                # unit vector point -> camera ... PO = P - Ow
    nrm = (Pw - Ow) / d[:, None]                                       # the reference's normal points camera -> point
    nrm = nrm + rng.normal(scale=0.6, size=nrm.shape)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    scale = 1.2 ** rng.uniform(-1, 9, n)
    maxd = (d * scale).astype(np.float32)                              # mfMaxDistance = dist * scaleFactor^level
    mind = (maxd / 1.2 ** 7).astype(np.float32)
    # edge cases
    k = min(n, 16)
    Pw[:k] = 0
    if k >= 4:
        Pw[0] = Ow                                                     # point at the camera centre: dist 0, z 0
        maxd[1], mind[1] = 0.0, 0.0
        Pw[2] = np.nan
        maxd[3] = np.inf
    return Pw, nrm.astype(np.float32), maxd, mind
