"""Synthetic inputs for SearchByBoW: two feature sets over one synthetic vocabulary (set 1 = the KeyFrame's queries,
set 2 = the Frame / second KeyFrame), their FeatureVectors as CSR, visit / availability masks and keypoint angles."""
import numpy as np

import bow_cases


def feature_vector_from_nodes(node_of_feature):
    """DBoW2::FeatureVector::addFeature (FeatureVector.cpp:30-45) over features 0..n-1: node ids ascending,
    feature indices ascending inside a node.  Features with node < 0 are in no list."""
    node_of_feature = np.asarray(node_of_feature, np.int64)
    keep = np.nonzero(node_of_feature >= 0)[0]
    order = keep[np.argsort(node_of_feature[keep], kind="stable")]
    nodes, counts = np.unique(node_of_feature[keep], return_counts=True)
    begin = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
    return nodes.astype(np.int32), begin, order.astype(np.int32)


def make_pair(seed, n1=900, n2=1100, n_nodes=60, flip=25, dup_frac=0.1, mask_frac=0.2, shuffle_lists=False,
              far_frac=0.02):
    """Set 2: random descriptors in n_nodes clusters.  Set 1: noisy copies of set-2 descriptors (same node, so they
    can match), exact duplicates (ties in distance -> first-in-list rule), bit complements (distance 256) and
    features in nodes set 2 does not have."""
    rng = np.random.default_rng(seed)
    node2 = rng.integers(0, n_nodes, n2) * 3 + 7          # sparse, non-contiguous node ids
    desc2 = rng.integers(0, 256, (n2, 32), dtype=np.uint8)
    src = rng.integers(0, max(n2, 1), n1) if n2 else np.zeros(n1, np.int64)
    desc1 = bow_cases._flip_bits(rng, desc2[src], rng.integers(0, flip + 1, n1)) if n2 else \
        rng.integers(0, 256, (n1, 32), dtype=np.uint8)
    node1 = node2[src].copy() if n2 else rng.integers(0, n_nodes, n1) * 3 + 7
    if n2:
        dup = rng.random(n1) < dup_frac                      # exact copies: several queries want the same train
        desc1[dup] = desc2[src[dup]]
        far = rng.random(n1) < far_frac
        desc1[far] = ~desc2[src[far]]                        # complement: distance 256 to its source
    lost = rng.random(n1) < 0.1
    node1[lost] = rng.integers(0, n_nodes, int(lost.sum())) * 3 + 8   # ids set 2 never has
    node1[rng.random(n1) < 0.03] = -1                        # not in the feature vector at all
    if n2:
        # near-duplicates inside set 2 so that best and second best are close (ratio test both ways)
        twin = rng.integers(0, n2, n2 // 5)
        desc2[twin] = bow_cases._flip_bits(rng, desc2[(twin + 1) % n2], rng.integers(0, 6, len(twin)))
        node2[twin] = node2[(twin + 1) % n2]
    fv1 = feature_vector_from_nodes(node1)
    fv2 = feature_vector_from_nodes(node2)
    if shuffle_lists:                                        # list order is an input, not an invariant
        for fv in (fv1, fv2):
            for r in range(len(fv[0])):
                rng.shuffle(fv[2][fv[1][r]:fv[1][r + 1]])
    valid1 = (rng.random(n1) >= mask_frac).astype(np.uint8)
    avail2 = (rng.random(n2) >= mask_frac).astype(np.uint8)
    rot = rng.choice([10.0, 200.0, 355.0])
    angle2 = rng.uniform(0, 360, n2).astype(np.float32)
    angle1 = np.mod((angle2[src] if n2 else np.zeros(n1)) + rot + rng.normal(0, 4, n1), 360).astype(np.float32)
    wild = rng.random(n1) < 0.3
    angle1[wild] = rng.uniform(0, 360, int(wild.sum())).astype(np.float32)
    return dict(desc1=np.ascontiguousarray(desc1), desc2=np.ascontiguousarray(desc2), valid1=valid1, avail2=avail2,
                fv1=fv1, fv2=fv2, angle1=angle1, angle2=angle2)


def naive(p, th_low, inclusive, nnratio, check_orientation):
    """Definition-level restatement in plain Python/numpy (dict-based merge, explicit loops) used to pin the C++ oracle."""
    d1, d2 = p["desc1"], p["desc2"]
    n1, n2 = len(d1), len(d2)
    pop = np.array([bin(i).count("1") for i in range(256)])
    lists1 = {int(nd): p["fv1"][2][p["fv1"][1][r]:p["fv1"][1][r + 1]] for r, nd in enumerate(p["fv1"][0])}
    lists2 = {int(nd): p["fv2"][2][p["fv2"][1][r]:p["fv2"][1][r + 1]] for r, nd in enumerate(p["fv2"][0])}
    m12 = -np.ones(n1, np.int64)
    taken = np.zeros(n2, bool)
    hist = [[] for _ in range(30)]
    for nd in sorted(set(lists1) & set(lists2)):
        for i1 in lists1[nd]:
            if not p["valid1"][i1]:
                continue
            b1, b2, bi = 256, 256, -1
            for i2 in lists2[nd]:
                if taken[i2] or (p["avail2"] is not None and not p["avail2"][i2]):
                    continue
                d = int(pop[d1[i1] ^ d2[i2]].sum())
                if d < b1:
                    b2, b1, bi = b1, d, i2
                elif d < b2:
                    b2 = d
            if (b1 <= th_low if inclusive else b1 < th_low) and np.float32(b1) < np.float32(nnratio) * np.float32(b2):
                m12[i1] = bi
                taken[bi] = True
                if check_orientation:
                    rot = np.float32(p["angle1"][i1]) - np.float32(p["angle2"][bi])
                    if rot < 0:
                        rot = np.float32(rot + np.float32(360.0))
                    x = float(np.float32(rot * np.float32(1.0 / 30)))
                    b = int(np.floor(x + 0.5))           # round half away from zero for x >= 0
                    hist[0 if b == 30 else b].append(i1)
    if check_orientation:
        sizes = [len(h) for h in hist]
        mx = [0, 0, 0]
        ind = [-1, -1, -1]
        for i, s in enumerate(sizes):
            if s > mx[0]:
                mx = [s, mx[0], mx[1]]
                ind = [i, ind[0], ind[1]]
            elif s > mx[1]:
                mx = [mx[0], s, mx[1]]
                ind = [ind[0], i, ind[1]]
            elif s > mx[2]:
                mx[2] = s
                ind[2] = i
        if mx[1] < np.float32(0.1) * np.float32(mx[0]):
            ind[1] = ind[2] = -1
        elif mx[2] < np.float32(0.1) * np.float32(mx[0]):
            ind[2] = -1
        for i in range(30):
            if i not in ind:
                for i1 in hist[i]:
                    m12[i1] = -1
    return int((m12 >= 0).sum()), m12


KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"),
                     ("octave", "<i4"), ("class_id", "<i4")])


def make_triangulation_pair(seed, n1=900, n2=1000, n_nodes=40, pix_noise=1.5, mask_frac=0.3, flip=20):
    """Two pinhole KeyFrames looking at the same 3-D points: set 2's keypoints are projections of random points, set 1's
    are the same points seen from a camera moved by (R12, t12) plus pixel noise (so the epipolar gate passes for most
    true pairs and fails for most wrong ones), descriptors = noisy copies, same BoW node."""
    rng = np.random.default_rng(seed)
    fx, fy, cx, cy = 718.856, 718.856, 607.19, 185.21
    ang = rng.normal(0, 0.03, 3)
    Rx = np.array([[1, 0, 0], [0, np.cos(ang[0]), -np.sin(ang[0])], [0, np.sin(ang[0]), np.cos(ang[0])]])
    Ry = np.array([[np.cos(ang[1]), 0, np.sin(ang[1])], [0, 1, 0], [-np.sin(ang[1]), 0, np.cos(ang[1])]])
    Rz = np.array([[np.cos(ang[2]), -np.sin(ang[2]), 0], [np.sin(ang[2]), np.cos(ang[2]), 0], [0, 0, 1]])
    R12 = Rx @ Ry @ Rz
    t12 = np.array([0.3, 0.02, -0.9]) + rng.normal(0, 0.05, 3)      # mostly forward motion: the epipole is in the image
    K = np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1.0]])
    tx = np.array([[0, -t12[2], t12[1]], [t12[2], 0, -t12[0]], [-t12[1], t12[0], 0]])
    F12 = (np.linalg.inv(K.T) @ tx @ R12 @ np.linalg.inv(K)).astype(np.float32)
    C2 = -R12.T @ t12                                                # camera-1 centre in camera-2 coordinates
    ep = np.array([fx * C2[0] / C2[2] + cx, fy * C2[1] / C2[2] + cy], np.float32)
    z = rng.uniform(4, 40, n2)
    u2 = rng.uniform(0, 1241, n2)
    v2 = rng.uniform(0, 376, n2)
    near = rng.random(n2) < 0.05                                     # a few keypoints right at the epipole
    u2[near] = ep[0] + rng.normal(0, 6, int(near.sum()))
    v2[near] = ep[1] + rng.normal(0, 6, int(near.sum()))
    X2 = np.stack([(u2 - cx) / fx * z, (v2 - cy) / fy * z, z], 1)
    src = rng.integers(0, max(n2, 1), n1) if n2 else np.zeros(n1, np.int64)
    X1 = X2[src] @ R12.T + t12 if n2 else np.zeros((n1, 3)) + [0, 0, 5.0]
    u1 = fx * X1[:, 0] / X1[:, 2] + cx + rng.normal(0, pix_noise, n1)
    v1 = fy * X1[:, 1] / X1[:, 2] + cy + rng.normal(0, pix_noise, n1)
    node2 = rng.integers(0, n_nodes, n2) * 5 + 2
    desc2 = rng.integers(0, 256, (n2, 32), dtype=np.uint8)
    if n2:
        twin = rng.integers(0, n2, n2 // 4)                          # equal-distance candidates: last-minimum rule
        desc2[twin] = desc2[(twin + 1) % n2]
        node2[twin] = node2[(twin + 1) % n2]
    desc1 = bow_cases._flip_bits(rng, desc2[src], rng.integers(0, flip + 1, n1)) if n2 else \
        rng.integers(0, 256, (n1, 32), dtype=np.uint8)
    node1 = node2[src].copy() if n2 else rng.integers(0, n_nodes, n1) * 5 + 2
    node1[rng.random(n1) < 0.08] = 1
    kp1, kp2 = np.zeros(n1, KP_DTYPE), np.zeros(n2, KP_DTYPE)
    kp1["x"], kp1["y"], kp2["x"], kp2["y"] = u1, v1, u2, v2
    kp2["octave"] = rng.integers(0, 8, n2)
    kp1["octave"] = rng.integers(0, 8, n1)
    kp2["angle"] = rng.uniform(0, 360, n2)
    kp1["angle"] = np.mod((kp2["angle"][src] if n2 else 0) + 12.0 + rng.normal(0, 5, n1), 360)
    wild = rng.random(n1) < 0.25
    kp1["angle"][wild] = rng.uniform(0, 360, int(wild.sum()))
    scale = (np.float32(1.2) ** np.arange(8)).astype(np.float32)
    return dict(desc1=np.ascontiguousarray(desc1), desc2=np.ascontiguousarray(desc2),
                valid1=(rng.random(n1) >= mask_frac).astype(np.uint8), avail2=(rng.random(n2) >= mask_frac).astype(np.uint8),
                stereo1=(rng.random(n1) < 0.5).astype(np.uint8), stereo2=(rng.random(n2) < 0.5).astype(np.uint8),
                fv1=feature_vector_from_nodes(node1), fv2=feature_vector_from_nodes(node2), kp1=kp1, kp2=kp2,
                scale_factors2=scale, level_sigma2_2=(scale * scale).astype(np.float32), F12=F12, ep=ep)


def naive_triangulation(p, coarse, check_orientation):
    """Definition-level restatement of ORBmatcher.cc:1168-1402 in plain Python (float32 steps spelled with numpy scalars,
    fused products through float64, which is exact for one float32 fma)."""
    f32 = np.float32

    def fma(a, b, c):      # a*b exact in float64 (24+24 bits), + c rounds once to float64 then to float32: double rounding
        return f32(np.float64(a) * np.float64(b) + np.float64(c))   # is harmless here except in rare ties; see test note

    d1, d2 = p["desc1"], p["desc2"]
    n1, n2 = len(d1), len(d2)
    pop = np.array([bin(i).count("1") for i in range(256)])
    lists1 = {int(nd): p["fv1"][2][p["fv1"][1][r]:p["fv1"][1][r + 1]] for r, nd in enumerate(p["fv1"][0])}
    lists2 = {int(nd): p["fv2"][2][p["fv2"][1][r]:p["fv2"][1][r + 1]] for r, nd in enumerate(p["fv2"][0])}
    F = p["F12"].reshape(3, 3).astype(f32)
    ep = p["ep"].astype(f32)
    m12 = -np.ones(n1, np.int64)
    taken = np.zeros(n2, bool)
    hist = [[] for _ in range(30)]
    for nd in sorted(set(lists1) & set(lists2)):
        for i1 in lists1[nd]:
            if not p["valid1"][i1]:
                continue
            x1, y1 = f32(p["kp1"]["x"][i1]), f32(p["kp1"]["y"][i1])
            a = f32(fma(x1, F[0, 0], f32(y1 * F[1, 0])) + F[2, 0])
            b = f32(fma(x1, F[0, 1], f32(y1 * F[1, 1])) + F[2, 1])
            c = f32(fma(x1, F[0, 2], f32(y1 * F[1, 2])) + F[2, 2])
            best, bi = 50, -1
            for i2 in lists2[nd]:
                if taken[i2] or not p["avail2"][i2]:
                    continue
                d = int(pop[d1[i1] ^ d2[i2]].sum())
                if d > 50 or d > best:
                    continue
                x2, y2 = f32(p["kp2"]["x"][i2]), f32(p["kp2"]["y"][i2])
                if not p["stereo1"][i1] and not p["stereo2"][i2]:
                    ex, ey = f32(ep[0] - x2), f32(ep[1] - y2)
                    if fma(ex, ex, f32(ey * ey)) < f32(f32(100) * p["scale_factors2"][p["kp2"]["octave"][i2]]):
                        continue
                ok = bool(coarse)
                if not ok:
                    num = f32(fma(a, x2, f32(b * y2)) + c)
                    den = fma(a, a, f32(b * b))
                    if den != 0:
                        dsqr = f32(f32(num * num) / den)
                        ok = np.float64(dsqr) < 3.84 * np.float64(p["level_sigma2_2"][p["kp2"]["octave"][i2]])
                if ok:
                    best, bi = d, i2
            if bi >= 0:
                m12[i1] = bi
                taken[bi] = True
                if check_orientation:
                    rot = f32(p["kp1"]["angle"][i1]) - f32(p["kp2"]["angle"][bi])
                    if rot < 0:
                        rot = f32(rot + f32(360.0))
                    b_ = int(np.floor(float(f32(rot * f32(1.0 / 30))) + 0.5))
                    hist[0 if b_ == 30 else b_].append(i1)
    if check_orientation:
        mx, ind = [0, 0, 0], [-1, -1, -1]
        for i, h in enumerate(hist):
            s = len(h)
            if s > mx[0]:
                mx, ind = [s, mx[0], mx[1]], [i, ind[0], ind[1]]
            elif s > mx[1]:
                mx, ind = [mx[0], s, mx[1]], [ind[0], i, ind[1]]
            elif s > mx[2]:
                mx[2], ind[2] = s, i
        if mx[1] < f32(0.1) * f32(mx[0]):
            ind[1] = ind[2] = -1
        elif mx[2] < f32(0.1) * f32(mx[0]):
            ind[2] = -1
        for i in range(30):
            if i not in ind:
                for i1 in hist[i]:
                    m12[i1] = -1
    return int((m12 >= 0).sum()), m12
