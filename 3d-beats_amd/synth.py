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


def balanced_tree(k, max_depth, num_classes, calib, retries=6, device=None):
    """A tree whose deep levels are OCCUPIED: every node's feature is drawn from the reference's proposal distribution
    (decision_tree.py:353-367) and its threshold is the median of that feature over the calibration pixels that reach the
    node -- what a best-gini split of tree_train.cu:99-236 looks like to the evaluator: data-adapted and roughly
    balanced, not a random cut that sends most pixels one way (the "full" topology draws the threshold from the proposal
    distribution too, which leaves 98 % of a depth-20 tree's last level unreachable).  A feature that is constant over a
    node's pixels (both probes on the same cell) is redrawn, as a trainer would never pick it.  Nodes that fewer than two
    calibration pixels reach keep the proposal's own threshold.  Every walk reaches level D-1 (worst case for the
    memory system); leaf PDFs are integers/256 (order-independent sums).

    Level-synchronous: the calibration pixels stay grouped by node and one 64-bit sort per level (key = node | feature |
    pixel) orders every node's values at once.  The array work runs in torch on `device` (default: the GPU when there is
    one -- a depth-22 tree over 7 M pixels takes a second there and a minute on the host); integer arithmetic and IEEE
    fp64 divides only, so the tree does not depend on the device (tests/test_synth.py)."""
    import torch
    if device is None:
        device = "cuda" if torch.cuda.is_available() else "cpu"
    dev = torch.device(device)
    rng = np.random.default_rng(TREE_SEED_BASE + 100003 + int(k))
    n_nodes = (1 << max_depth) - 1
    t = np.zeros((n_nodes, 7 + 2 * num_classes), dtype=np.float32)
    n_img, h, w = calib.shape
    cal = torch.from_numpy(np.ascontiguousarray(calib).astype(np.int32)).to(dev).reshape(-1).to(torch.int64)
    pix = torch.nonzero((cal != 0) & (cal != NO_PIXEL)).reshape(-1)     # flat positions of the pixels a forest evaluates (tree_eval.cu:88-89)
    npx = int(pix.numel())
    pix_bits = max(1, int(max(npx - 1, 1)).bit_length())
    assert max_depth - 1 + 18 + pix_bits <= 63, "too many calibration pixels for the 64-bit sort key"
    p_base = (pix // (h * w)) * (h * w)
    p_y = (pix // w) % h
    p_x = pix % w
    p_d = cal[pix].to(torch.float64)

    def feature_values(sel, uv):
        """depth[u] - depth[v] (decision_tree_common.hpp:8-28, scale 1) of the pixels `sel` for one (ux, uy, vx, vy) row each"""
        d, x, y, b = p_d[sel], p_x[sel], p_y[sel], p_base[sel]

        def probe(ox, oy):
            px = x + torch.floor(ox / d).to(torch.int64)
            py = y + torch.floor(oy / d).to(torch.int64)
            ok = (px >= 0) & (px < w) & (py >= 0) & (py < h)
            v = cal[torch.where(ok, b + py * w + px, torch.zeros_like(px))]
            return torch.where(ok, v, torch.full_like(v, NO_PIXEL))

        return probe(uv[:, 0], uv[:, 1]) - probe(uv[:, 2], uv[:, 3])

    def seg_sum(flags, starts, counts):
        """per node: sum of flags over its run [start, start + count)"""
        cs = torch.cat([torch.zeros(1, dtype=torch.int64, device=dev), torch.cumsum(flags.to(torch.int64), 0)])
        return cs[starts + counts] - cs[starts]

    order = torch.arange(npx, dtype=torch.int64, device=dev)      # pixel ranks grouped by node of the current level, heap order
    counts = torch.tensor([npx], dtype=torch.int64, device=dev)
    for j in range(max_depth):
        cnt = 1 << j
        base = cnt - 1
        feats_np = _features(rng, cnt)
        feats = torch.from_numpy(feats_np.astype(np.float64)).to(dev)
        node_ids = torch.arange(cnt, dtype=torch.int64, device=dev)
        node_of = torch.repeat_interleave(node_ids, counts)
        starts = torch.cumsum(counts, 0) - counts
        f = feature_values(order, feats[node_of, 0:4])
        for _ in range(retries):
            # nodes (of two pixels or more) on which the feature takes one value only: redraw
            change = torch.zeros(npx, dtype=torch.bool, device=dev)
            if npx > 1:
                change[1:] = (f[1:] != f[:-1]) & (node_of[1:] == node_of[:-1])
            bad = (counts >= 2) & (seg_sum(change, starts, counts) == 0)
            n_bad = int(bad.sum())
            if n_bad == 0:
                break
            redraw = _features(rng, n_bad)
            bad_np = bad.cpu().numpy()
            feats_np[bad_np] = redraw
            feats[bad] = torch.from_numpy(redraw.astype(np.float64)).to(dev)
            sel = bad[node_of]
            f[sel] = feature_values(order[sel], feats[node_of[sel], 0:4])
        key, _ = torch.sort((node_of << (18 + pix_bits)) | ((f + 65536) << pix_bits) | order)
        order = key & ((1 << pix_bits) - 1)
        fs = ((key >> pix_bits) & 0x3FFFF) - 65536
        # threshold T (f < T goes left): the cut nearest to the middle of the node's sorted values
        fm = fs[torch.clamp(starts + counts // 2, max=max(npx - 1, 0))] if npx else torch.zeros(cnt, dtype=torch.int64, device=dev)
        node_key = node_ids << (18 + pix_bits)
        below = torch.searchsorted(key, node_key | ((fm + 65536) << pix_bits)) - starts        # values < fm
        upto = torch.searchsorted(key, node_key | ((fm + 65537) << pix_bits)) - starts         # values <= fm
        take_upto = torch.abs(2 * upto - counts) < torch.abs(2 * below - counts)
        left = torch.where(take_upto, upto, below)
        T = torch.where(take_upto, fm + 1, fm).to(torch.float64) - 0.5
        use = (counts >= 2) & (left > 0) & (left < counts)
        thr = torch.where(use, T, feats[:, 4])
        # nodes the calibration set cannot place: the proposal's own threshold decides where their pixels go
        rest = ~use & (counts > 0)
        if bool(rest.any()):
            go_left = (fs.to(torch.float64) < thr[node_of]) & rest[node_of]
            left = torch.where(rest, seg_sum(go_left, starts, counts), left)
        left = torch.where(counts > 0, left, torch.zeros_like(left))
        feats_np[:, 4] = thr.cpu().numpy().astype(np.float32)
        t[base:base + cnt, 0:5] = feats_np
        if j < max_depth - 1:
            t[base:base + cnt, 5:7] = -1.0
            counts = torch.stack([left, counts - left], dim=1).reshape(-1)
        else:
            t[base:base + cnt, 7:] = rng.integers(0, 257, size=(cnt, 2 * num_classes)).astype(np.float32) / 256.0
    return t


def calibration_frames(n=16, h=480, w=848, first_idx=9000, max_depth=20):
    """Frames a balanced forest takes its medians from, seeds away from every evaluated batch: the bench batch's mix (half
    dense, half live-like: 3.7 M pixels, seven per node of a depth-20 tree's last level); deeper trees get dense frames
    (depth 21: 6.5 M pixels, depth 22 and more: 13 M), so that their last levels are occupied as well."""
    if max_depth <= 20:
        return mixed_batch(n, first_idx, h, w)
    return frames(["dense"] * (n if max_depth == 21 else 2 * n), first_idx, h, w)


def forest(num_trees, max_depth, num_classes, topology="full", first_tree=0, calib=None, device=None):
    """topology: "full", "trained" (trained-like) or "balanced" (median thresholds over `calib`, default
    calibration_frames(); `device`: where balanced_tree does its array work)."""
    f = np.empty((num_trees, (1 << max_depth) - 1, 7 + 2 * num_classes), dtype=np.float32)
    if topology == "balanced":
        calib = calibration_frames(max_depth=max_depth) if calib is None else calib
        for k in range(num_trees):
            f[k] = balanced_tree(first_tree + k, max_depth, num_classes, calib, device=device)
        return f
    mk = full_tree if topology == "full" else trained_like_tree
    for k in range(num_trees):
        f[k] = mk(first_tree + k, max_depth, num_classes)
    return f


def algorithmic_bytes(n_frames, h, w, labels_reduce, with_filter, n_classes, stats):
    """sum_frames[2HW + 2HlWl (+2HlWl filter)] + 32 * node records read + 4C * leaves reached."""
    hl, wl = h // labels_reduce, w // labels_reduce
    b = n_frames * (2 * h * w + 2 * hl * wl + (2 * hl * wl if with_filter else 0))
    return int(b + 32 * int(stats[1]) + 4 * n_classes * int(stats[2]))
