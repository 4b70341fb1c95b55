"""Synthetic local maps for the tracking front-end chain (BASELINE configs[2]): map points that project
near the frame's keypoints under a given pose, with descriptors that are noisy copies of the keypoints' — what
Tracking::SearchLocalPoints (Tracking.cc:3343-3388) sees — plus the oracle composition the device chain must reproduce."""
import numpy as np

import frustum_cases as fc
import matcher_cases as mc


def local_map(seed, kps, desc, ur, depth, R, t, Ow, scale, M, copy_frac=0.7, bad_frac=0.03, spars_frac=0.05,
              obs_zero_frac=0.15, skip_frac=0.1):
    """M local map points in WORLD coordinates for the camera pose (R, t, Ow): 70 % sit on the viewing ray of a keypoint
    (a few pixels off), at the keypoint's stereo depth when it has one, with a descriptor a few bits away; the rest are
    anywhere in front of / around the camera with random descriptors."""
    rng = np.random.Generator(np.random.PCG64(seed))
    c = fc.KITTI_CAM
    n = len(kps)
    nlev = len(scale)
    src = rng.integers(0, max(n, 1), M)
    is_copy = (rng.random(M) < copy_frac) & (n > 0)
    u = np.where(is_copy, kps["x"][src] + rng.normal(0, 2.0, M), rng.uniform(-100, c["bounds"][1] + 100, M))
    v = np.where(is_copy, kps["y"][src] + rng.normal(0, 2.0, M), rng.uniform(-60, c["bounds"][3] + 60, M))
    z = np.where(is_copy & (depth[src] > 0), depth[src] * rng.uniform(0.97, 1.03, M), rng.uniform(2.0, 70.0, M))
    Pc = np.stack([(u - c["cx"]) * z / c["fx"], (v - c["cy"]) * z / c["fy"], z], 1)
    Pw = ((Pc - t.astype(np.float64)) @ R.astype(np.float64)).astype(np.float32)
    po = Pw.astype(np.float64) - Ow.astype(np.float64)
    d = np.linalg.norm(po, axis=1) + 1e-9
    nrm = po / d[:, None] + rng.normal(scale=0.35, size=po.shape)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    lvl = np.where(is_copy, kps["octave"][src] + rng.integers(0, 2, M), rng.integers(0, nlev, M)).clip(0, nlev - 1)
    # mfMaxDistance = dist * scaleFactor^level at creation (MapPoint.cc:505-513); jitter keeps PredictScale near `lvl`
    maxd = (d * np.asarray(scale, np.float64)[lvl] * rng.uniform(0.93, 1.0, M)).astype(np.float32)
    mind = (maxd / np.float32(scale[-1])).astype(np.float32)
    dsc = np.where(is_copy[:, None], mc.flip_bits(rng, desc[src] if n else np.zeros((M, 32), np.uint8), 40),
                   rng.integers(0, 256, (M, 32), dtype=np.uint8)).astype(np.uint8)
    return dict(pos_w=Pw, normal=nrm.astype(np.float32), max_distance=maxd, min_distance=mind,
                visit=(rng.random(M) >= skip_frac).astype(np.uint8), bad=(rng.random(M) < bad_frac).astype(np.uint8),
                sparsified=(rng.random(M) < spars_frac).astype(np.uint8), desc=dsc,
                obs=np.where(rng.random(M) < obs_zero_frac, 0, rng.integers(1, 12, M)).astype(np.int32))


def oracle_local_points(oracle, oframe, frustum, mp, frame_mp, th, bFar=False, thFar=50.0, nnratio=0.8, cos_limit=0.5):
    """The reference's composition on the CPU oracle: isInFrustum for the visited points (Frame.cc:512-571), then
    SearchByProjection over the whole table (ORBmatcher.cc:43-142).  -> (nmatches, scratch dict); frame_mp in place."""
    r = oracle.is_in_frustum(frustum, mp["pos_w"], mp["normal"], mp["max_distance"], mp["min_distance"], cos_limit)
    visit = mp["visit"].astype(bool) if mp.get("visit") is not None else np.ones(len(mp["obs"]), bool)
    tab = dict(track_in_view=(r["track_in_view"].astype(bool) & visit).astype(np.uint8), bad=mp["bad"], sparsified=mp["sparsified"],
               proj_x=r["proj_x"], proj_y=r["proj_y"], proj_xr=r["proj_xr"], track_depth=r["track_depth"], level=r["level"],
               view_cos=r["view_cos"], desc=mp["desc"], obs=mp["obs"])
    nm = oframe.SearchByProjection_mps(tab, frame_mp, th, bFar, thFar, nnratio)
    return nm, r, visit


def oracle_topk(oracle, oframe, frustum, mp, th, scale, k=8, bFar=False, thFar=50.0, cos_limit=0.5):
    """Per map point the k best candidates (index, distance) of the window search against a frame that holds no map
    points, in the reference's scan order (ties -> earlier in GetFeaturesInArea's order): ORBmatcher.cc:52-120 on the oracle."""
    r = oracle.is_in_frustum(frustum, mp["pos_w"], mp["normal"], mp["max_distance"], mp["min_distance"], cos_limit)
    m = len(mp["obs"])
    flags = mp["flags"] if "flags" in mp else (mp["visit"].astype(np.uint8) | (mp["bad"].astype(np.uint8) << 1) | (mp["sparsified"].astype(np.uint8) << 2))
    idx = np.full((m, k), -1, np.int32)
    dist = np.full((m, k), 256, np.int32)
    ur = np.asarray(oframe.u_right, np.float32)
    for i in range(m):
        if not (flags[i] & 1) or not r["track_in_view"][i] or (flags[i] & 2):
            continue
        if bFar and r["track_depth"][i] > thFar:
            continue
        lvl = int(r["level"][i])
        rad = np.float32(2.5 if np.float64(r["view_cos"][i]) > 0.998 else 4.0)
        if th != 1.0:
            rad = np.float32(rad * np.float32(th))
        rr = np.float32(rad * np.float32(scale[lvl]))
        cand = oframe.GetFeaturesInArea(float(r["proj_x"][i]), float(r["proj_y"][i]), float(rr), lvl - 1, lvl)
        keep = [j for j in cand if not (ur[j] > 0 and abs(np.float32(r["proj_xr"][i]) - ur[j]) > rr)]
        if not keep:
            continue
        dd = np.array([oracle.descriptor_distance(mp["desc"][i], oframe.desc[j]) for j in keep], np.int32)
        order = np.argsort(dd, kind="stable")[:k]
        idx[i, :len(order)] = np.asarray(keep, np.int32)[order]
        dist[i, :len(order)] = dd[order]
    return idx, dist, r

