"""Seeded synthetic depth frames and forests (the reference ships no data or models).

Frame and forest distributions are the ones SURVEY.md section 8(d) fixes:
  * frames: numpy default_rng(20211003 + frame index); "dense" and "live-like" uint16 frames;
  * forests: per node u,v = (cos t, sin t) * e^U(0,14), thresh = +-e^U(0,11) -- the reference's own
    proposal distribution (/root/reference/src/decision_tree.py:353-367) -- seeds 777 + tree index;
    "full" and "trained-like" topologies; node record layout as written by the reference trainer
    (/root/reference/src/cuda/tree_train.cu:183-235): [ux,uy,vx,vy,thresh,l_next,r_next,l_pdf[C],r_pdf[C]].
Pure numpy; used by tests, bench.py and __graft_entry__.smoke().
"""
import numpy as np

FRAME_SEED_BASE = 20211003
TREE_SEED_BASE = 777
FEATURE_MAGNITUDE_MAX = 14.0  # decision_tree.py:353
FEATURE_THRESHOLD_MAX = 11.0  # decision_tree.py:354
NO_PIXEL = 65535


def _surface(rng, h, w):
    """N(4000,400) per pixel plus a smooth low-frequency field (4 random 2-D cosines, total amplitude 300)."""
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    z = rng.normal(4000.0, 400.0, size=(h, w))
    for _ in range(4):
        fx, fy = rng.uniform(0.2, 3.0, size=2) * 2 * np.pi / max(h, w)
        ph = rng.uniform(0, 2 * np.pi)
        z += 75.0 * np.cos(fx * xx + fy * yy + ph)
    return np.clip(np.rint(z), 1, 65534).astype(np.uint16)


def dense_frame(idx, h=480, w=848):
    """Every pixel valid."""
    rng = np.random.default_rng(FRAME_SEED_BASE + int(idx))
    return _surface(rng, h, w)


def live_frame(idx, h=480, w=848):
    """Background 65535, one filled ellipse (~15 % of the frame) of surface, 0.5 % zero-valued holes."""
    rng = np.random.default_rng(FRAME_SEED_BASE + int(idx))
    z = _surface(rng, h, w)
    ratio = rng.uniform(1.0, 1.5)
    a = np.sqrt(0.15 * h * w * ratio / np.pi)
    b = a / ratio
    cx, cy = rng.uniform(0.35, 0.65) * w, rng.uniform(0.35, 0.65) * h
    th = rng.uniform(0, np.pi)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    xr = (xx - cx) * np.cos(th) + (yy - cy) * np.sin(th)
    yr = -(xx - cx) * np.sin(th) + (yy - cy) * np.cos(th)
    inside = (xr / a) ** 2 + (yr / b) ** 2 <= 1.0
    out = np.full((h, w), NO_PIXEL, dtype=np.uint16)
    out[inside] = z[inside]
    holes = rng.random((h, w)) < 0.005
    out[holes] = 0
    return out


def frames(kinds, first_idx=0, h=480, w=848):
    """kinds: iterable of 'dense' / 'live'; frame i uses seed FRAME_SEED_BASE + first_idx + i."""
    out = np.empty((len(kinds), h, w), dtype=np.uint16)
    for i, k in enumerate(kinds):
        out[i] = (dense_frame if k == "dense" else live_frame)(first_idx + i, h, w)
    return out


def mixed_batch(n, first_idx=0, h=480, w=848):
    """Half dense, half live-like (config 4's 512 + 512 split applied per shard), alternating, so that
    every contiguous chunk of the batch carries the same mix."""
    kinds = ["dense" if i % 2 == 0 else "live" for i in range(n)]
    return frames(kinds, first_idx, h, w)


def _features(rng, n):
    def offs():
        th = rng.uniform(0, 2 * np.pi, n)
        mag = np.exp(rng.uniform(0, FEATURE_MAGNITUDE_MAX, n))
        return np.cos(th) * mag, np.sin(th) * mag

    ux, uy = offs()
    vx, vy = offs()
    thr = rng.choice([-1.0, 1.0], n) * np.exp(rng.uniform(0, FEATURE_THRESHOLD_MAX, n))
    return np.stack([ux, uy, vx, vy, thr], axis=1).astype(np.float32)


def full_tree(k, max_depth, num_classes):
    """Worst case: every walk reaches level D-1; leaf PDFs are integers/256 (order-independent sums)."""
    rng = np.random.default_rng(TREE_SEED_BASE + int(k))
    n = (1 << max_depth) - 1
    t = np.zeros((n, 7 + 2 * num_classes), dtype=np.float32)
    t[:, 0:5] = _features(rng, n)
    first_last = (1 << (max_depth - 1)) - 1
    t[:first_last, 5:7] = -1.0
    n_last = n - first_last
    t[first_last:, 7:] = rng.integers(0, 257, size=(n_last, 2 * num_classes)).astype(np.float32) / 256.0
    return t


def trained_like_tree(k, max_depth, num_classes, leaf_prob=0.15, min_leaf_level=3):
    """Each side turns into a leaf with prob. leaf_prob from level min_leaf_level on; unreachable nodes
    stay all-zero (decision_tree.py:446); PDFs are normalised random counts (tree_train.cu:201-207)."""
    rng = np.random.default_rng(TREE_SEED_BASE + int(k))
    n = (1 << max_depth) - 1
    C = num_classes
    t = np.zeros((n, 7 + 2 * C), dtype=np.float32)
    reach = np.array([True])
    for j in range(max_depth):
        base = (1 << j) - 1
        cnt = 1 << j
        idx = base + np.nonzero(reach)[0]
        m = idx.shape[0]
        t[idx, 0:5] = _features(rng, m)
        last = j == max_depth - 1
        if last:
            leaf = np.ones((m, 2), dtype=bool)
        elif j < min_leaf_level:
            leaf = np.zeros((m, 2), dtype=bool)
        else:
            leaf = rng.random((m, 2)) < leaf_prob
        t[idx, 5] = np.where(leaf[:, 0], 0.0, -1.0)
        t[idx, 6] = np.where(leaf[:, 1], 0.0, -1.0)
        counts = rng.integers(0, 1000, size=(m, 2, C)).astype(np.float32)
        counts[:, :, 0] += 1.0
        pdf = counts / counts.sum(axis=2, keepdims=True, dtype=np.float32)
        pdf = np.where(leaf[:, :, None], pdf, 0.0).astype(np.float32)
        t[idx, 7:] = pdf.reshape(m, 2 * C)
        if not last:
            nxt = np.zeros(2 * cnt, dtype=bool)
            g = np.nonzero(reach)[0]
            nxt[2 * g] = ~leaf[:, 0]
            nxt[2 * g + 1] = ~leaf[:, 1]
            reach = nxt
    return t


def forest(num_trees, max_depth, num_classes, topology="full", first_tree=0):
    mk = full_tree if topology == "full" else trained_like_tree
    f = np.empty((num_trees, (1 << max_depth) - 1, 7 + 2 * num_classes), dtype=np.float32)
    for k in range(num_trees):
        f[k] = mk(first_tree + k, max_depth, num_classes)
    return f


def algorithmic_bytes(n_frames, h, w, labels_reduce, with_filter, n_classes, stats):
    """sum_frames[2HW + 2HlWl (+2HlWl filter)] + 32 * node records read + 4C * leaves reached."""
    hl, wl = h // labels_reduce, w // labels_reduce
    b = n_frames * (2 * h * w + 2 * hl * wl + (2 * hl * wl if with_filter else 0))
    return int(b + 32 * int(stats[1]) + 4 * n_classes * int(stats[2]))
