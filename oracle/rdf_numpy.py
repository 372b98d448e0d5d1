"""Independent numpy restatement (level-synchronous) of the reference RDF kernels.

TEST INFRASTRUCTURE ONLY (see rdf_oracle.c).  Written separately from the C oracle so that the two
can be checked against each other; also timed by bench.py as the "numpy fallback" CPU baseline
the north_star alludes to (the reference itself has no CPU path).  Parity unpinned by reference
fixtures.

Follows /root/reference/src/cuda/tree_eval.cu:24-137 (forest), :140-212 (tree), :214-248
(composite), :7-21 (argmax), decision_tree_common.hpp:8-28 (feature), cu_utils.hpp:19-40,79-86.
"""
import numpy as np

NO_PIXEL = 65535


def _floor_sat_i32(q):
    """__float2int_rd on a float32 array: floor, saturate, NaN -> 0.  Returns int64 holding int32 values."""
    with np.errstate(invalid="ignore"):
        f = np.floor(q.astype(np.float32)).astype(np.float64)
    f = np.where(np.isnan(f), 0.0, f)
    f = np.clip(f, -2147483648.0, 2147483647.0)
    return f.astype(np.int64)


def _wrap_i32(a):
    return ((a + 2**31) % 2**32) - 2**31


def _gather_depth(frames, img, y, x):
    """Array3d.get with per-axis bounds check and default 65535 (cu_utils.hpp:58-62,79-86)."""
    n, h, w = frames.shape
    ok = (x >= 0) & (x < w) & (y >= 0) & (y < h)
    out = np.full(x.shape, NO_PIXEL, dtype=np.uint16)
    out[ok] = frames[img[ok], y[ok], x[ok]]
    return out


def _walk_tree(frames, tree, D, C, img, y, x, d, s):
    """Returns (leaf_pdf [P,C] float32, reached [P] bool, levels [P] int)."""
    P = img.shape[0]
    g = np.zeros(P, dtype=np.int64)
    active = np.ones(P, dtype=bool)
    reached = np.zeros(P, dtype=bool)
    leaf = np.zeros((P, C), dtype=np.float32)
    levels = np.zeros(P, dtype=np.int64)
    df = d.astype(np.float32)
    s = np.float32(s)
    for j in range(D):
        if not active.any():
            break
        a = np.nonzero(active)[0]
        node = tree[(1 << j) - 1 + g[a]]  # [A, E]
        levels[a] += 1
        with np.errstate(all="ignore"):
            qux = (s * node[:, 0]).astype(np.float32) / df[a]
            quy = (s * node[:, 1]).astype(np.float32) / df[a]
            qvx = (s * node[:, 2]).astype(np.float32) / df[a]
            qvy = (s * node[:, 3]).astype(np.float32) / df[a]
        ux = _wrap_i32(x[a] + _floor_sat_i32(qux))
        uy = _wrap_i32(y[a] + _floor_sat_i32(quy))
        vx = _wrap_i32(x[a] + _floor_sat_i32(qvx))
        vy = _wrap_i32(y[a] + _floor_sat_i32(qvy))
        du = _gather_depth(frames, img[a], uy, ux).astype(np.float32)
        dv = _gather_depth(frames, img[a], vy, vx).astype(np.float32)
        f = du - dv
        with np.errstate(invalid="ignore"):
            left = f < node[:, 4]
        l_next = _floor_sat_i32(node[:, 5])
        r_next = _floor_sat_i32(node[:, 6])
        cont = np.where(left, l_next == -1, r_next == -1)
        # leaves
        lf = ~cont
        if lf.any():
            idx = a[lf]
            side = np.where(left[lf], 0, 1)
            pdfs = node[lf][:, 7:].reshape(-1, 2, C)
            leaf[idx] = pdfs[np.arange(idx.shape[0]), side]
            reached[idx] = True
            active[idx] = False
        # descend
        ci = a[cont]
        g[ci] = g[ci] * 2 + np.where(left[cont], 0, 1)
    return leaf, reached, levels


def _argmax(pdf):
    """First index with the largest value strictly > 0, else 0; NaN never wins (tree_eval.cu:7-21)."""
    P, C = pdf.shape
    best = np.zeros(P, dtype=np.float32)
    cls = np.zeros(P, dtype=np.int64)
    for j in range(C):
        with np.errstate(invalid="ignore"):
            m = pdf[:, j] > best
        best = np.where(m, pdf[:, j], best)
        cls = np.where(m, j, cls)
    return cls


def eval_forest(depth, forest, labels_out, labels_reduce=1, filter_images=None, filter_class=None,
                scale_factor=1.0, stats=None):
    depth = np.ascontiguousarray(depth, dtype=np.uint16)
    forest = np.ascontiguousarray(forest, dtype=np.float32)
    n, h, w = depth.shape
    T, n_nodes, E = forest.shape
    D = int(np.log2(n_nodes + 1))
    C = (E - 7) // 2
    r = int(labels_reduce)
    lh, lw = h // r, w // r
    assert labels_out.shape == (n, lh, lw)
    img, ly, lx = np.meshgrid(np.arange(n), np.arange(lh), np.arange(lw), indexing="ij")
    img, ly, lx = img.ravel(), ly.ravel(), lx.ravel()
    keep = np.ones(img.shape, dtype=bool)
    if filter_images is not None:
        keep &= filter_images[img, ly, lx].astype(np.int64) == int(filter_class)
    y, x = ly * r, lx * r
    d = _gather_depth(depth, img, y, x)
    keep &= (d != 0) & (d != NO_PIXEL)
    img, ly, lx, y, x, d = img[keep], ly[keep], lx[keep], y[keep], x[keep], d[keep]
    pdf = np.zeros((img.shape[0], C), dtype=np.float32)
    tot_lv = tot_lf = 0
    for k in range(T):
        leaf, reached, lv = _walk_tree(depth, forest[k], D, C, img, y, x, d, scale_factor)
        with np.errstate(all="ignore"):
            pdf[reached] = pdf[reached] + leaf[reached]
        tot_lv += int(lv.sum())
        tot_lf += int(reached.sum())
    labels_out[img, ly, lx] = _argmax(pdf).astype(np.uint16)
    if stats is not None:
        stats[0] += img.shape[0]
        stats[1] += tot_lv
        stats[2] += tot_lf
    return labels_out


def eval_tree(depth, tree, labels_out):
    depth = np.ascontiguousarray(depth, dtype=np.uint16)
    tree = np.ascontiguousarray(tree, dtype=np.float32)
    n, h, w = depth.shape
    n_nodes, E = tree.shape
    D = int(np.log2(n_nodes + 1))
    C = (E - 7) // 2
    img, y, x = np.meshgrid(np.arange(n), np.arange(h), np.arange(w), indexing="ij")
    img, y, x = img.ravel(), y.ravel(), x.ravel()
    d = depth[img, y, x]
    keep = (d != 0) & (d != NO_PIXEL)
    img, y, x, d = img[keep], y[keep], x[keep], d[keep]
    leaf, reached, _ = _walk_tree(depth, tree, D, C, img, y, x, d, 1.0)
    lab = _argmax(leaf[reached]).astype(np.uint16)
    labels_out[img[reached], y[reached], x[reached]] = lab
    return labels_out


def composite(label_images, conditions, out):
    cond = np.asarray(conditions, dtype=np.int32).reshape(-1, 2)
    imgs = [np.asarray(a, dtype=np.uint16) for a in label_images]
    h, w = imgs[0].shape[-2:]
    o = out.reshape(h, w)
    bad = 0
    for yy in range(h):
        for xx in range(w):
            off = 0
            for i, im in enumerate(imgs):
                l = int(im.reshape(h, w)[yy, xx])
                if l == 0 or l == NO_PIXEL:
                    break
                e = off + l - 1
                if e < 0 or e >= cond.shape[0]:
                    bad += 1
                    break
                t, v = int(cond[e, 0]), int(cond[e, 1])
                if t == 0:
                    o[yy, xx] = np.uint16(v & 0xFFFF)
                    break
                off = v
            else:
                bad += 1
    return bad
