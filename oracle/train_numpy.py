"""numpy restatement of 3d-beats' decision-tree trainer.  TEST INFRASTRUCTURE ONLY.

Follows /root/reference/src/decision_tree.py:444-600 (DecisionTreeTrainer.train: level-synchronous loop
over proposal blocks, node blocks and image blocks) and /root/reference/src/cuda/tree_train.cu:
  evaluate_random_features :4-64     per (pixel, proposal): count the pixel's label in the child it falls into
  gini helpers             :66-97    fp32 arithmetic on uint64 counts
  pick_best_features       :99-236   best proposal per active node, node record, child counts
  get_active_nodes_next_level :238-273, copy_pixel_groups :275-324
and decision_tree.py:353-371 (random proposals from the global numpy RNG).

Parity unpinned by reference fixtures.  Two places where this restatement has to *define* the semantics:
  * fp32 expressions are evaluated as written, one rounding per operation (nvcc may contract `p += p_i*p_i`
    into an fma, so near-tied gains can rank differently on the reference's own hardware);
  * the reference appends next-level nodes with an atomic counter, i.e. in scheduler order; the order has no
    effect on the trained tree, and here it is ascending.
Stale PDF entries left behind when a later proposal block overwrites a node (tree_train.cu:204-223) are
reproduced: nothing is cleared.
"""
import numpy as np

FEATURE_MAGNITUDE_MAX = 14.
FEATURE_THRESHOLD_MAX = 11.
F32 = np.float32


def make_random_features(n):
    """decision_tree.py:353-371, same draws from the global numpy RNG in the same order."""
    out = []
    for _ in range(n):
        offs = []
        for _ in range(2):
            th = np.random.uniform(0, np.pi * 2)
            mag = np.power(np.e, np.random.uniform(0, FEATURE_MAGNITUDE_MAX))
            offs.append(np.array([np.cos(th), np.sin(th)]) * mag)
        thr = np.random.choice([-1, 1]) * np.power(np.e, np.random.uniform(0, FEATURE_THRESHOLD_MAX))
        out.append((offs[0][0], offs[0][1], offs[1][0], offs[1][1], thr))
    return np.array(out, dtype=np.float32)


def _floor_sat_i32(q):
    with np.errstate(invalid="ignore"):
        f = np.floor(q.astype(np.float32)).astype(np.float64)
    f = np.where(np.isnan(f), 0.0, f)
    return np.clip(f, -2147483648.0, 2147483647.0).astype(np.int64)


def _wrap(a):
    return ((a + 2**31) % 2**32) - 2**31


def compute_feature(depth, img, y, x, u, v):
    """decision_tree_common.hpp:8-28 with uv_scale = 1 for pixel arrays (img, y, x) and one feature (u, v)."""
    n, h, w = depth.shape
    d = depth[img, y, x]
    df = d.astype(np.float32)
    with np.errstate(all="ignore"):
        ox = _floor_sat_i32(F32(u[0]) / df); oy = _floor_sat_i32(F32(u[1]) / df)
        px = _floor_sat_i32(F32(v[0]) / df); py = _floor_sat_i32(F32(v[1]) / df)

    def get(yy, xx):
        ok = (xx >= 0) & (xx < w) & (yy >= 0) & (yy < h)
        out = np.full(xx.shape, 65535, dtype=np.uint16)
        out[ok] = depth[img[ok], yy[ok], xx[ok]]
        return out.astype(np.float32)

    f = get(_wrap(y + oy), _wrap(x + ox)) - get(_wrap(y + py), _wrap(x + px))
    return np.where(d == 0, F32(0), f).astype(np.float32)


def _gini_impurity(c):
    s = F32(np.uint64(c.sum()))
    p = F32(0)
    for ci in c:
        with np.errstate(all="ignore"):
            p_i = F32(np.uint64(ci)) / s
            p = F32(p + F32(p_i * p_i))
    return F32(F32(1) - p)


def _gini_gain(pc, lc, rc):
    with np.errstate(all="ignore"):
        p_sum = F32(np.uint64(pc.sum()))
        p_imp = _gini_impurity(pc)
        rem = F32(F32(F32(F32(np.uint64(lc.sum())) / p_sum) * _gini_impurity(lc)) +
                  F32(F32(F32(np.uint64(rc.sum())) / p_sum) * _gini_impurity(rc)))
        return F32(p_imp - rem)


def _cutoff(counts, total, thresh=F32(0.999)):
    for i, c in enumerate(counts):
        with np.errstate(all="ignore"):
            if F32(F32(np.uint64(c)) * F32(1)) / F32(np.uint64(total)) >= thresh:
                return i
    return -1


def train_tree(depth, labels, num_classes, max_depth, proposal_blocks_per_level, proposals_per_block,
               max_next_nodes_per_block=1 << 17, proposal_fn=make_random_features):
    """Returns the trained tree, float32 [2^D - 1, 7 + 2C] (level-order, tree_train.cu record layout)."""
    depth = np.ascontiguousarray(depth, dtype=np.uint16)
    labels = np.ascontiguousarray(labels, dtype=np.uint16)
    C, D = num_classes, max_depth
    E = 7 + 2 * C
    tree = np.zeros(((1 << D) - 1, E), dtype=np.float32)
    img, yy, xx = np.nonzero(labels > 0)
    lab = labels[img, yy, xx].astype(np.int64)
    node = np.zeros(img.shape, dtype=np.int64)           # nodes_by_pixel of the labelled pixels; -1 = retired
    max_leaf = 1 << D
    node_counts = np.zeros((max_leaf, C), dtype=np.uint64)
    np.add.at(node_counts[0], lab, 1)
    next_counts = node_counts.copy()                     # cu_array.to_gpu(self.node_counts) twice (:399-400)
    active = np.array([0], dtype=np.int64)
    for level in range(D):
        if active.size == 0:
            break
        best_gain = np.full(active.size, -1.0, dtype=np.float32)
        level_base = (1 << level) - 1
        for _ in range(proposal_blocks_per_level):
            props = proposal_fn(proposals_per_block)
            P = props.shape[0]
            max_next = 1 << (level + 1)
            if max_next > max_next_nodes_per_block:
                blocks = [(i * max_next_nodes_per_block, (i + 1) * max_next_nodes_per_block)
                          for i in range(max_next // max_next_nodes_per_block)]
            else:
                blocks = [(0, max_next)]
            for start, end in blocks:
                # ---- evaluate_random_features: counts[j][child - start][label] ----
                live = node >= 0
                elig = live & (node * 2 >= start) & (node * 2 + 1 < end)
                counts = np.zeros((P, min(max_next, max_next_nodes_per_block), C), dtype=np.uint64)
                ii, ey, ex, en, el = img[elig], yy[elig], xx[elig], node[elig], lab[elig]
                for j in range(P):
                    f = compute_feature(depth, ii, ey, ex, props[j, 0:2], props[j, 2:4])
                    with np.errstate(invalid="ignore"):
                        child = np.where(f < props[j, 4], en * 2, en * 2 + 1) - start
                    np.add.at(counts[j], (child, el), 1)
                # ---- pick_best_features ----
                for i, parent in enumerate(active):
                    l_child, r_child = parent * 2, parent * 2 + 1
                    if l_child < start or r_child >= end:
                        continue
                    pc = node_counts[parent]
                    p_sum = int(pc.sum())
                    best_g, best_j = F32(-1), 0
                    for j in range(P):
                        lc, rc = counts[j, l_child - start], counts[j, r_child - start]
                        ls, rs = int(lc.sum()), int(rc.sum())
                        assert ls + rs == p_sum
                        g = F32(0) if (ls == 0 or rs == 0) else _gini_gain(pc, lc, rc)
                        if g > best_g:
                            best_g, best_j = g, j
                    if not best_g > best_gain[i]:
                        continue
                    best_gain[i] = best_g
                    lc, rc = counts[best_j, l_child - start], counts[best_j, r_child - start]
                    ls, rs = int(lc.sum()), int(rc.sum())
                    rec = tree[level_base + parent]
                    rec[0:5] = props[best_j]
                    if best_g <= 0:
                        rec[5] = rec[6] = 0.0
                        for k in range(C):
                            with np.errstate(all="ignore"):
                                p = F32(F32(np.uint64(pc[k])) * F32(1)) / F32(np.uint64(p_sum))
                            rec[7 + k] = rec[7 + C + k] = p
                        continue
                    for side, cc, cs, child in ((0, lc, ls, l_child), (1, rc, rs, r_child)):
                        cut = _cutoff(cc, cs)
                        if cut > -1:
                            rec[5 + side] = 0.0
                            rec[7 + side * C + cut] = 1.0
                        elif level == D - 1:
                            rec[5 + side] = 0.0
                            for k in range(C):
                                with np.errstate(all="ignore"):
                                    rec[7 + side * C + k] = F32(F32(np.uint64(cc[k])) * F32(1)) / F32(np.uint64(cs))
                        else:
                            rec[5 + side] = -1.0
                            next_counts[child] = cc
        # ---- get_active_nodes_next_level (ascending order here) ----
        nxt = []
        for parent in active:
            rec = tree[level_base + parent]
            if rec[5] == -1.0:
                nxt.append(parent * 2)
            if rec[6] == -1.0:
                nxt.append(parent * 2 + 1)
        if level == D - 1:
            break
        node_counts = next_counts.copy()
        # ---- copy_pixel_groups ----
        cur = node.copy()
        idx = np.nonzero(cur >= 0)[0]
        for parent in np.unique(cur[idx]):
            sel = idx[cur[idx] == parent]
            rec = tree[level_base + parent]
            f = compute_feature(depth, img[sel], yy[sel], xx[sel], rec[0:2], rec[2:4])
            with np.errstate(invalid="ignore"):
                left = f < rec[4]
            status = np.where(left, _floor_sat_i32(np.array([rec[5]]))[0], _floor_sat_i32(np.array([rec[6]]))[0])
            node[sel] = np.where(status != -1, -1, parent * 2 + np.where(left, 0, 1))
        active = np.array(nxt, dtype=np.int64)
    return tree
