"""Seeded synthetic DBoW2-style vocabularies and map-point observation sets (shared by CPU and GPU tests and bench).

The real ORBvoc.txt (k=10, L=6, ~1.08 M nodes, 145 MB) is not in the reference checkout (.MISSING_LARGE_BLOBS), so
trees of the same shape are generated: children are noisy copies of their parent (hierarchical clusters), leaves
carry idf-like weights, a fraction of the words is "stopped" (weight 0), and `irregular` trees end some branches
early and vary the branching factor, as k-means trees do."""
import numpy as np


def _flip_bits(rng, desc, nbits):
    """desc [n,32] uint8 -> copy with `nbits` random bit flips per row (nbits scalar or [n])."""
    out = desc.copy()
    n = len(out)
    nb = np.broadcast_to(np.asarray(nbits), (n,))
    for t in range(int(nb.max()) if n else 0):
        rows = np.nonzero(nb > t)[0]
        pos = rng.integers(0, 256, len(rows))
        out[rows, pos // 8] ^= (1 << (pos % 8)).astype(np.uint8)
    return out


def make_vocabulary(seed, k=10, L=3, irregular=False, stop_frac=0.05, tie_frac=0.0, dfs_ids=False, flip=40):
    """-> dict(k, L, parent[int32 n], is_leaf[uint8 n], descriptors[uint8 n,32], weights[f64 n]); node 0 = root."""
    rng = np.random.default_rng(seed)
    parent = [np.zeros(1, np.int64)]
    desc = [rng.integers(0, 256, (1, 32), dtype=np.uint8)]
    level_of = [np.zeros(1, np.int64)]
    first = 0          # id of the first node of the current level
    cur = np.array([0])
    n_total = 1
    for lvl in range(1, L + 1):
        if irregular and lvl > 1:
            cnt = rng.integers(0, k + 1, len(cur))       # 0 children -> that node stays a (shallow) leaf
            cnt[rng.random(len(cur)) < 0.5] = k
        else:
            cnt = np.full(len(cur), k)
        if cnt.sum() == 0:
            cnt[0] = k
        par = np.repeat(cur, cnt)
        src = np.concatenate(desc)[par]
        child = _flip_bits(rng, src, max(2, flip >> (lvl - 1)))
        if tie_frac > 0:   # duplicate the previous sibling's descriptor: exercises the first-minimum tie break
            dup = np.nonzero((rng.random(len(par)) < tie_frac) & (np.arange(len(par)) > 0))[0]
            dup = dup[par[dup] == par[dup - 1]]
            child[dup] = child[dup - 1]
        ids = np.arange(n_total, n_total + len(par))
        parent.append(par)
        desc.append(child)
        level_of.append(np.full(len(par), lvl))
        cur = ids
        n_total += len(par)
    parent = np.concatenate(parent).astype(np.int32)
    desc = np.concatenate(desc)
    n = len(parent)
    has_child = np.zeros(n, bool)
    has_child[parent[1:]] = True
    is_leaf = (~has_child).astype(np.uint8)
    is_leaf[0] = 0
    w = np.zeros(n, np.float64)
    nl = int(is_leaf.sum())
    w[is_leaf > 0] = -np.log(rng.uniform(1e-4, 0.9, nl))      # idf = -log(Ni/N)
    stopped = np.nonzero(is_leaf)[0][rng.random(nl) < stop_frac]
    w[stopped] = 0.0
    if dfs_ids:            # renumber in depth-first order (DBoW2 creates nodes recursively); parent < child holds
        kids = [[] for _ in range(n)]
        for i in range(1, n):
            kids[parent[i]].append(i)
        order, stack = [], [0]
        while stack:
            x = stack.pop()
            order.append(x)
            stack.extend(reversed(kids[x]))
        new_id = np.empty(n, np.int64)
        new_id[order] = np.arange(n)
        perm = np.array(order)
        parent = new_id[parent[perm]].astype(np.int32)
        parent[0] = 0
        desc, is_leaf, w = desc[perm], is_leaf[perm], w[perm]
    return dict(k=k, L=L, parent=parent, is_leaf=is_leaf, descriptors=np.ascontiguousarray(desc), weights=w)


def make_features(seed, voc, n, noise=12):
    """n descriptors: noisy copies of random leaves (so descents are decisive) mixed with 20 % pure noise."""
    rng = np.random.default_rng(seed)
    leaves = np.nonzero(voc["is_leaf"])[0]
    pick = leaves[rng.integers(0, len(leaves), n)]
    d = _flip_bits(rng, voc["descriptors"][pick], rng.integers(0, noise + 1, n))
    rnd = rng.random(n) < 0.2
    d[rnd] = rng.integers(0, 256, (int(rnd.sum()), 32), dtype=np.uint8)
    return np.ascontiguousarray(d)


def write_text(path, voc, scoring=0, weighting=0, trailing_newline=True):
    """ORBvoc.txt layout, as TemplatedVocabulary::saveToTextFile writes it (TemplatedVocabulary.h:1427-1450)."""
    with open(path, "w") as f:
        f.write(f"{voc['k']} {voc['L']}  {scoring} {weighting}\n")
        n = len(voc["parent"])
        for i in range(1, n):
            d = " ".join(str(int(b)) for b in voc["descriptors"][i])
            f.write(f"{voc['parent'][i]} {int(voc['is_leaf'][i])} {d}  {float(voc['weights'][i])!r}")
            if i < n - 1 or trailing_newline:
                f.write("\n")


def make_observations(seed, sizes):
    """Map points with the given numbers of observed descriptors: noisy copies of one descriptor per point, with
    exact duplicates sprinkled in (equal medians -> first-minimum rule).  -> (desc [sum,32], obs_begin)."""
    rng = np.random.default_rng(seed)
    sizes = np.asarray(sizes, np.int64)
    ob = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    out = np.zeros((int(ob[-1]), 32), np.uint8)
    for p, n in enumerate(sizes):
        if n == 0:
            continue
        base = rng.integers(0, 256, (1, 32), dtype=np.uint8)
        d = _flip_bits(rng, np.repeat(base, n, axis=0), rng.integers(0, 60, n))
        if n > 2:
            dup = rng.integers(0, n, max(1, n // 4))
            d[dup] = d[(dup + 1) % n]
        out[ob[p]:ob[p + 1]] = d
    return out, ob
