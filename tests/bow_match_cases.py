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
