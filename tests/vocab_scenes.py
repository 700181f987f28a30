"""Synthetic DBoW2 vocabularies in the ORBvoc.txt format (no vocabulary file ships with the reference and there is no network)."""
import numpy as np


def make_vocabulary(rng, k, L, scoring=0, weighting=0, stop_frac=0.03, ragged=False, min_leaf_level=1):
    """Random k-ary tree of depth L in depth-first line order (like DBoW2's saveToTextFile).  Children are noisy copies of their
    parent so that descents are decided by few bits and ties do occur.  ragged: some branches end above level L (never above
    `min_leaf_level`).  Returns (header, parent[], is_leaf[], desc[n,32], weight[n])."""
    parent, leaf, desc, weight = [], [], [], []

    def add(pid, pdesc, level):
        d = pdesc.copy()
        for b in rng.integers(0, 256, int(rng.integers(4, 40))):
            d[b >> 3] ^= np.uint8(1 << (b & 7))
        parent.append(pid); desc.append(d)
        nid = len(parent)
        is_leaf = level == L or (ragged and level >= min_leaf_level and rng.random() < 0.15)
        leaf.append(1 if is_leaf else 0)
        if is_leaf:
            weight.append(0.0 if rng.random() < stop_frac else float(rng.uniform(0.2, 9.0)))
        else:
            weight.append(0.0)
            for _ in range(k if not ragged else int(rng.integers(2, k + 1))):
                add(nid, d, level + 1)

    root = rng.integers(0, 256, 32, dtype=np.uint8)
    for _ in range(k):
        add(0, root, 1)
    return (k, L, scoring, weighting), np.array(parent, np.int32), np.array(leaf, np.uint8), np.array(desc, np.uint8), np.array(weight, np.float64)


def write_text(path, header, parent, leaf, desc, weight):
    lines = ["%d %d %d %d" % header]
    for p, l, d, w in zip(parent, leaf, desc, weight):
        lines.append("%d %d %s %s" % (p, l, " ".join(str(int(x)) for x in d), repr(float(w))))
    with open(path, "w") as f:
        f.write("\n".join(lines))          # no trailing newline: the reference's reader would turn an empty last line into a node


def descriptors_near(rng, desc, n, maxflips=60):
    """n query descriptors: noisy copies of random vocabulary nodes (so every level has a meaningful nearest child)."""
    src = desc[rng.integers(0, len(desc), n)].copy()
    for r in range(n):
        for b in rng.integers(0, 256, int(rng.integers(0, maxflips + 1))):
            src[r, b >> 3] ^= np.uint8(1 << (b & 7))
    return src


def make_vocabulary_fast(rng, k, L, stop_frac=0.03):
    """The same kind of tree, full (every branch reaches level L), generated level by level with numpy (breadth-first line order: the reader only
    needs a parent to come before its children) - for ORBvoc-sized trees (k = 10, L = 6: 1 111 110 nodes, the size of the shipped ORBvoc.txt)."""
    parents, descs, leaves, weights = [], [], [], []
    root = rng.integers(0, 256, 32, dtype=np.uint8)
    prev_desc = root[None, :]; prev_ids = np.zeros(1, np.int64); next_id = 1
    for level in range(1, L + 1):
        n = len(prev_desc) * k
        d = np.repeat(prev_desc, k, axis=0)
        flips = rng.integers(4, 40, n)
        for r in range(int(flips.max())):                     # one random bit per round in the rows that still have flips left
            rows = np.nonzero(flips > r)[0]
            bits = rng.integers(0, 256, len(rows))
            d[rows, bits >> 3] ^= (1 << (bits & 7)).astype(np.uint8)
        ids = np.arange(next_id, next_id + n, dtype=np.int64); next_id += n
        parents.append(np.repeat(prev_ids, k)); descs.append(d)
        is_leaf = level == L
        leaves.append(np.full(n, 1 if is_leaf else 0, np.uint8))
        w = np.zeros(n, np.float64)
        if is_leaf:
            w = rng.uniform(0.2, 9.0, n); w[rng.random(n) < stop_frac] = 0.0
        weights.append(w)
        prev_desc, prev_ids = d, ids
    return (k, L, 0, 0), np.concatenate(parents).astype(np.int32), np.concatenate(leaves), np.concatenate(descs), np.concatenate(weights)


def write_text_fast(path, header, parent, leaf, desc, weight):
    """write_text for a million nodes: the 32 descriptor bytes of a line through one lookup table of decimal strings"""
    lut = np.array([str(i) for i in range(256)], dtype=object)
    with open(path, "w") as f:
        f.write("%d %d %d %d" % header)
        n = len(parent)
        for a in range(0, n, 50000):
            b = min(a + 50000, n)
            ds = lut[desc[a:b]]
            rows = ["\n%d %d %s %s" % (parent[i], leaf[i], " ".join(ds[i - a]), repr(float(weight[i]))) for i in range(a, b)]
            f.write("".join(rows))
