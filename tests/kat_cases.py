"""Hand-derived known-answer cases for the RDF kernels.

Every expected value below was worked out BY HAND from the text of the reference kernels
(/root/reference/src/cuda/tree_eval.cu, decision_tree_common.hpp, cu_utils.hpp) -- not by running
any implementation.  They pin the oracle (tests/test_oracle_kat.py) and, through the same table,
the HIP path (tests/test_gpu_parity.py).  `U` marks a pixel the kernel must leave untouched.

A node record is [ux, uy, vx, vy, thresh, l_next, r_next, l_pdf[C], r_pdf[C]]
(tree_train.cu:183-235); a flag of -1 means "descend", anything whose floor is not -1 means "leaf".
"""
import numpy as np

U = "U"  # untouched


def node(u=(0, 0), v=(0, 0), thresh=0.0, l_next=0.0, r_next=0.0, l_pdf=(), r_pdf=()):
    assert len(l_pdf) == len(r_pdf)
    return [u[0], u[1], v[0], v[1], thresh, l_next, r_next] + list(l_pdf) + list(r_pdf)


def tree(nodes):
    t = np.array(nodes, dtype=np.float32)
    n = t.shape[0]
    assert (n + 1) & n == 0, "complete tree needs 2^D - 1 nodes"
    return t


def onehot(c, C):
    p = [0.0] * C
    p[c] = 1.0
    return p


def _f(name, cite, depth, trees, expected, **kw):
    depth = np.array(depth, dtype=np.uint16)
    if depth.ndim == 2:
        depth = depth[None]
    forest = np.stack([tree(t) for t in trees])
    case = dict(kind="forest", name=name, cite=cite, depth=depth, forest=forest, expected=expected,
                labels_reduce=1, scale_factor=1.0, filter=None, filter_class=None)
    case.update(kw)
    return case


def _t(name, cite, depth, tr, expected):
    depth = np.array(depth, dtype=np.uint16)
    if depth.ndim == 2:
        depth = depth[None]
    return dict(kind="tree", name=name, cite=cite, depth=depth, tree=tree(tr), expected=expected)


def expected_array(expected, prefill):
    """Turn a nested list with U markers into a uint16 array, U -> prefill."""
    e = np.array([[[prefill if v == U else v for v in row] for row in img] for img in expected], dtype=np.uint16)
    return e


def cases():
    C3 = 3
    out = []
    L1, R2 = onehot(1, C3), onehot(2, C3)

    # 1. root is a leaf on both sides; u = v = 0 so f = d - d = 0 (decision_tree_common.hpp:24-27).
    #    f < thresh picks the left PDF, otherwise the right one (tree_eval.cu:107-121).
    flat = [[1000, 1000], [1000, 1000]]
    for thr, lab in [(1.0, 1), (-1.0, 2), (0.0, 2)]:  # 0 < 0 is false -> right
        out.append(_f(f"root_leaf_thresh_{thr}", "tree_eval.cu:107-121,132-135",
                      flat, [[node(thresh=thr, l_pdf=[0.1, 0.7, 0.2], r_pdf=[0.2, 0.3, 0.5])]],
                      [[[lab, lab], [lab, lab]]]))

    # 2. centre depth 0 or 65535 returns before any write (tree_eval.cu:88-89): output keeps its pre-fill.
    out.append(_f("invalid_centre_untouched", "tree_eval.cu:88-89",
                  [[0, 65535], [1000, 5]], [[node(thresh=1.0, l_pdf=L1, r_pdf=R2)]],
                  [[[U, U], [1, 1]]]))

    # 3. out-of-bounds probes read 65535, checked per axis (cu_utils.hpp:79-86).  All depths are 1, so the
    #    pixel offset equals floor(u).  In bounds: f = 1 - 1 = 0 < 60000 -> left(1); out of bounds:
    #    f = 65535 - 1 >= 60000 -> right(2).  A flat-index implementation would wrap into the next row.
    ones = [[1, 1, 1]] * 3
    for name, u, exp in [
        ("x_plus", (1, 0), [[1, 1, 2]] * 3),
        ("x_minus", (-1, 0), [[2, 1, 1]] * 3),
        ("y_plus", (0, 1), [[1, 1, 1], [1, 1, 1], [2, 2, 2]]),
        ("y_minus", (0, -1), [[2, 2, 2], [1, 1, 1], [1, 1, 1]]),
    ]:
        out.append(_f(f"oob_{name}", "cu_utils.hpp:79-86; decision_tree_common.hpp:15-25",
                      ones, [[node(u=u, thresh=60000.0, l_pdf=L1, r_pdf=R2)]], [exp]))
    #    batch of two images: the last row of image 0 must not read image 1, nor image 1 past the end.
    out.append(_f("oob_y_plus_batch", "cu_utils.hpp:79-86; tree_eval.cu:64-67",
                  [ones, ones], [[node(u=(0, 1), thresh=60000.0, l_pdf=L1, r_pdf=R2)]],
                  [[[1, 1, 1], [1, 1, 1], [2, 2, 2]]] * 2))

    # 4. __float2int_rd floors toward -inf (decision_tree_common.hpp:15-18): u.x = -1.5, d = 1 -> dx = -2.
    #    frame [100,1,1,1], thresh 1000, v = 0:
    #    x=0: d=100, dx=floor(-0.015)=-1 -> OOB 65535-100 -> right(2)
    #    x=1: dx=-2 -> x=-1 OOB -> 65534 -> right(2)   (truncation would read x=0: 100-1=99 -> left)
    #    x=2: dx=-2 -> x=0 -> 100-1=99 < 1000 -> left(1);  x=3: x=1 -> 1-1=0 -> left(1)
    out.append(_f("floor_negative_offset", "decision_tree_common.hpp:15-18",
                  [[100, 1, 1, 1]], [[node(u=(-1.5, 0), thresh=1000.0, l_pdf=L1, r_pdf=R2)]],
                  [[[2, 2, 1, 1]]]))

    # 5. scale_factor multiplies the offset before the divide (decision_tree_common.hpp:16): u.x = 4.
    #    frame [1,1,100,1,1,1], thresh 50.  s=0.5: dx=2 (x=2 has d=100 -> dx=0);  s=1: dx=4.
    fr5 = [[1, 1, 100, 1, 1, 1]]
    n5 = [[node(u=(4, 0), thresh=50.0, l_pdf=L1, r_pdf=R2)]]
    out.append(_f("scale_half", "decision_tree_common.hpp:8-19", fr5, n5, [[[2, 1, 1, 1, 2, 2]]], scale_factor=0.5))
    out.append(_f("scale_one", "decision_tree_common.hpp:8-19", fr5, n5, [[[1, 1, 1, 2, 2, 2]]], scale_factor=1.0))

    # 6. labels_reduce: label (lx,ly) reads depth (r*lx, r*ly); label dims are integer-divided (tree_eval.cu:45,69-70).
    fr6 = np.full((5, 5), 65535, dtype=np.uint16)
    fr6[0, 0], fr6[0, 2], fr6[2, 0], fr6[2, 2], fr6[1, 1], fr6[4, 4] = 10, 0, 65535, 20, 30, 40
    out.append(_f("reduce2_odd_dims", "tree_eval.cu:45,64-70",
                  fr6, [[node(thresh=1.0, l_pdf=L1, r_pdf=R2)]], [[[1, U], [U, 1]]], labels_reduce=2))
    #    probes stay in full-resolution depth coordinates: label lx=1 sits at x=2 and probes x=3.
    out.append(_f("reduce2_probe_coords", "tree_eval.cu:69-70; decision_tree_common.hpp:15-25",
                  [[1, 9, 1, 100, 1, 1], [1, 1, 1, 1, 1, 1]],
                  [[node(u=(1, 0), thresh=50.0, l_pdf=L1, r_pdf=R2)]], [[[1, 2, 1]]], labels_reduce=2))

    # 7. filter image: a pixel is evaluated only where filter == filter_class (tree_eval.cu:81-85).
    out.append(_f("filter_class", "tree_eval.cu:81-85",
                  flat, [[node(thresh=1.0, l_pdf=L1, r_pdf=R2)]], [[[1, U], [U, 1]]],
                  filter=np.array([[[3, 2], [65535, 3]]], dtype=np.uint16), filter_class=3))

    # 8. argmax: strict >, starts at 0.f, first index wins, NaN never wins (tree_eval.cu:7-21).
    for name, pdf, lab in [
        ("tie_lowest_index", [0.25, 0.5, 0.5, 0.125], 1),
        ("all_zero", [0.0, 0.0, 0.0, 0.0], 0),
        ("all_nonpositive", [-1.0, -2.0, 0.0, -3.0], 0),
        ("nan_never_wins", [float("nan"), 0.3, float("nan"), 0.2], 1),
    ]:
        out.append(_f(f"argmax_{name}", "tree_eval.cu:7-21",
                      [[7]], [[node(thresh=1.0, l_pdf=pdf, r_pdf=[9, 9, 9, 9])]], [[[lab]]]))
    #    two trees: PDFs are summed per class before the argmax (tree_eval.cu:123-126).
    #    [0.125,0.25] + [0.5,0.125] = [0.625,0.375] -> 0 although tree 0 alone says 1.
    out.append(_f("two_tree_sum", "tree_eval.cu:123-135", [[7]],
                  [[node(thresh=1.0, l_pdf=[0.125, 0.25], r_pdf=[9, 9])],
                   [node(thresh=1.0, l_pdf=[0.5, 0.125], r_pdf=[9, 9])]], [[[0]]]))

    # 9. an all-zero (untrained) node: f = 0 < 0 is false -> right; flag 0 -> leaf of zeros -> label 0
    #    (decision_tree.py:446; tree_eval.cu:107-121).
    zero3 = [[0.0] * (7 + 2 * C3)] * 3
    out.append(_f("untrained_forest_writes_0", "tree_eval.cu:95-135", [[5, 6]], [zero3], [[[0, 0]]]))
    out.append(_t("untrained_tree_writes_0", "tree_eval.cu:176-209", [[5, 6]], zero3, [[[0, 0]]]))

    # 10. "descend" on the last level: g is updated, the loop ends, nothing is added (tree_eval.cu:95-128).
    fall = [node(thresh=1.0, l_next=-1.0, r_next=-1.0, l_pdf=L1, r_pdf=R2)]
    out.append(_f("fall_off_forest_writes_0", "tree_eval.cu:95-135", [[5]], [fall], [[[0]]]))
    out.append(_t("fall_off_tree_untouched", "tree_eval.cu:176-211", [[5]], fall, [[[U]]]))
    out.append(_f("fall_off_one_of_two", "tree_eval.cu:95-135", [[5]],
                  [fall, [node(thresh=1.0, l_pdf=[0.0, 0.25, 0.75], r_pdf=R2)]], [[[2]]]))

    # 11. flags go through __float2int_rd: [-1,0) floors to -1 = descend; -1.5 -> -2 and 0.5 -> 0 are leaves
    #     (tree_eval.cu:101-102).  Root sends everything left (f = 0 < 1).
    def t11(flag):
        return [node(thresh=1.0, l_next=flag, r_next=0.0, l_pdf=L1, r_pdf=L1),
                node(thresh=1.0, l_pdf=R2, r_pdf=R2), node(thresh=1.0, l_pdf=R2, r_pdf=R2)]
    for flag, lab in [(-0.5, 2), (-1.0, 2), (-1.5, 1), (0.5, 1), (-0.0, 1)]:
        out.append(_f(f"flag_floor_{flag}", "tree_eval.cu:101-102,107-121", [[5]], [t11(flag)], [[[lab]]]))

    # 13. a probe that lands on a 0-valued pixel uses 0.0, not "missing" (decision_tree_common.hpp:24-25).
    #     x=0: d=5, u.x=5 -> dx=1 -> depth 0 -> f = 0 - 5 = -5 < -1 -> left(1).  x=1 is invalid.
    out.append(_f("probe_on_zero_pixel", "decision_tree_common.hpp:24-25", [[5, 0]],
                  [[node(u=(5, 0), thresh=-1.0, l_pdf=L1, r_pdf=R2)]], [[[1, U]]]))

    # 15. image index decode in a batch (tree_eval.cu:64-67).
    out.append(_f("batch_decode", "tree_eval.cu:64-67", [[[1000, 0]], [[65535, 1000]]],
                  [[node(thresh=1.0, l_pdf=L1, r_pdf=R2)]], [[[1, U]], [[U, 1]]]))

    # 16. __float2int_rd saturates and maps NaN to 0; x + INT_MAX wraps negative -> out of bounds.
    #     u.x = +-3e10 (beyond int32 even after /7) or inf: probe OOB -> f = 65535 - 7 -> right(2).  u.x = NaN: offset 0 -> f = 0 -> left(1).
    for name, ux, lab in [("pos_huge", 3e10, 2), ("neg_huge", -3e10, 2), ("inf", float("inf"), 2),
                          ("neg_inf", float("-inf"), 2), ("nan", float("nan"), 1)]:
        out.append(_f(f"offset_{name}", "decision_tree_common.hpp:15-18 (__float2int_rd)", [[7, 7, 7]],
                      [[node(u=(ux, 0), thresh=50.0, l_pdf=L1, r_pdf=R2)]], [[[lab, lab, lab]]]))
        out.append(_f(f"offset_y_{name}", "decision_tree_common.hpp:15-18 (__float2int_rd)", [[7], [7], [7]],
                      [[node(u=(0, ux), thresh=50.0, l_pdf=L1, r_pdf=R2)]], [[[lab], [lab], [lab]]]))

    # 17. level-order addressing: children of node g on level j are 2g, 2g+1 on level j+1 (cu_utils.hpp:32-39).
    #     frame [1,1,100,1]; root u.x=1, thresh 50: x=0 -> 0 -> left child (node 1); x=1 -> 99 -> right child
    #     (node 2); x=2 (d=100, dx=0) -> left; x=3 -> OOB -> right.  node 1 answers 1, node 2 answers 3.
    C4 = 4
    t17 = [node(u=(1, 0), thresh=50.0, l_next=-1.0, r_next=-1.0, l_pdf=[0] * C4, r_pdf=[0] * C4),
           node(thresh=1.0, l_pdf=onehot(1, C4), r_pdf=onehot(2, C4)),
           node(thresh=-1.0, l_pdf=onehot(0, C4), r_pdf=onehot(3, C4))]
    out.append(_f("two_levels_heap_order", "cu_utils.hpp:32-39; tree_eval.cu:95-121",
                  [[1, 1, 100, 1]], [t17], [[[1, 3, 1, 3]]]))
    out.append(_t("two_levels_heap_order_tree", "cu_utils.hpp:32-39; tree_eval.cu:176-209",
                  [[1, 1, 100, 1]], t17, [[[1, 3, 1, 3]]]))

    # 18. the v probe is subtracted from the u probe (decision_tree_common.hpp:24-27): frame [3,10], d=3 at x=0:
    #     u.x=3 -> dx=1 -> 10; v=0 -> 3; f = 7.  thresh 7 -> 7 < 7 false -> right(2); thresh 7.5 -> left(1).
    #     Swapped (v.x=3): f = 3 - 10 = -7 < 7 -> left(1).  x=1: d=10, dx=floor(0.3)=0 -> f=0 -> left(1) always.
    out.append(_f("u_minus_v_right", "decision_tree_common.hpp:24-27", [[3, 10]],
                  [[node(u=(3, 0), thresh=7.0, l_pdf=L1, r_pdf=R2)]], [[[2, 1]]]))
    out.append(_f("u_minus_v_left", "decision_tree_common.hpp:24-27", [[3, 10]],
                  [[node(u=(3, 0), thresh=7.5, l_pdf=L1, r_pdf=R2)]], [[[1, 1]]]))
    out.append(_f("v_minus_swapped", "decision_tree_common.hpp:24-27", [[3, 10]],
                  [[node(v=(3, 0), thresh=7.0, l_pdf=L1, r_pdf=R2)]], [[[1, 1]]]))
    return out


def composite_cases():
    """14. The worked example the reference itself holds (decision_tree.py:214-220):
         conditions = [(0,1),(0,2),(1,3),(0,3),(0,4)]      i0 i1 | ID:  1 - | 1;  2 - | 2;  3 1 | 3;  3 2 | 4
       plus label 0 / 65535 in any consulted layer -> pixel untouched (tree_eval.cu:235)."""
    cond = np.array([(0, 1), (0, 2), (1, 3), (0, 3), (0, 4)], dtype=np.int32)
    i0 = np.array([[1, 2, 3, 3, 0, 65535, 3, 3]], dtype=np.uint16)
    i1 = np.array([[9, 9, 1, 2, 1, 1, 0, 65535]], dtype=np.uint16)
    return [dict(name="reference_worked_example", cite="decision_tree.py:214-220; tree_eval.cu:232-244",
                 images=[i0, i1], cond=cond, expected=[[[1, 2, 3, 4, U, U, U, U]]], bad=0),
            # single layer, all rows terminal
            dict(name="single_layer", cite="tree_eval.cu:232-240",
                 images=[np.array([[1, 2, 0], [2, 1, 65535]], dtype=np.uint16)],
                 cond=np.array([(0, 7), (0, 5)], dtype=np.int32), expected=[[[7, 5, U], [5, 7, U]]], bad=0),
            # a walk that never terminates (type 1 on the last layer) or leaves the table: the reference
            # device-asserts (tree_eval.cu:246-247); this build leaves the pixel untouched and counts it.
            dict(name="invalid_walks_counted", cite="tree_eval.cu:241-247",
                 images=[np.array([[1, 2, 3]], dtype=np.uint16)],
                 cond=np.array([(0, 4), (1, 0)], dtype=np.int32), expected=[[[4, U, U]]], bad=2)]
