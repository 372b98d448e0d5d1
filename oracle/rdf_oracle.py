"""ctypes front-end of oracle/librdf_oracle.so (the C restatement in rdf_oracle.c).

TEST INFRASTRUCTURE ONLY -- see the header of rdf_oracle.c.  Importable from tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg; the shipped package never imports it.
Parity status: *parity unpinned* by reference fixtures (the reference has none for this path).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "librdf_oracle.so")
_lib = None


def build(force=False):
    """Compile rdf_oracle.c with gcc (oracle/Makefile)."""
    src = os.path.join(_HERE, "rdf_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "librdf_oracle.so"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = ctypes.CDLL(_SO)
        i, f, p = ctypes.c_int, ctypes.c_float, ctypes.c_void_p
        L.rdf_oracle_eval_forest.argtypes = [p, i, i, i, p, i, i, i, p, i, p, i, f, p, i]
        L.rdf_oracle_eval_forest.restype = i
        L.rdf_oracle_eval_tree.argtypes = [p, i, i, i, p, i, i, p, i]
        L.rdf_oracle_eval_tree.restype = i
        L.rdf_oracle_composite.argtypes = [p, i, i, i, p, i, p, p]
        L.rdf_oracle_composite.restype = i
        L.rdf_oracle_order_sensitive.argtypes = [p, i, i, i, p, i, i, i, i, f]
        L.rdf_oracle_order_sensitive.restype = ctypes.c_int64
        L.rdf_oracle_max_threads.restype = i
        L.rdf_oracle_visit_map.argtypes = [p, i, i, i, p, i, i, i, i, f, p, i]
        L.rdf_oracle_visit_map.restype = i
        L.rdf_oracle_walk_lengths.argtypes = [p, i, i, i, p, i, i, i, i, f, p, i]
        L.rdf_oracle_walk_lengths.restype = i
        _lib = L
    return _lib


def _c(a, dtype):
    a = np.ascontiguousarray(a, dtype=dtype)
    return a


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _forest_dims(forest):
    assert forest.ndim == 3 and forest.dtype == np.float32
    T, n_nodes, E = forest.shape
    D = int(np.log2(n_nodes + 1))
    assert (1 << D) - 1 == n_nodes, "forest.shape[1] must be 2^D - 1"
    assert E >= 7 and (E - 7) % 2 == 0
    return T, D, (E - 7) // 2


def eval_forest(depth, forest, labels_out, labels_reduce=1, filter_images=None, filter_class=None,
                scale_factor=1.0, stats=None, n_threads=0):
    """labels_out (uint16 [N,H/r,W/r]) is updated in place; untouched pixels keep their value.

    stats: optional np.uint64[3] accumulator (evaluated pixels, node records read, leaves reached).
    n_threads: 0 = all cores.
    """
    depth = _c(depth, np.uint16)
    forest = _c(forest, np.float32)
    assert depth.ndim == 3
    n, h, w = depth.shape
    T, D, C = _forest_dims(forest)
    assert labels_out.dtype == np.uint16 and labels_out.flags["C_CONTIGUOUS"]
    assert labels_out.shape == (n, h // labels_reduce, w // labels_reduce)
    if filter_images is not None:
        assert filter_class is not None
        filter_images = _c(filter_images, np.uint16)
        assert filter_images.shape == labels_out.shape
        fptr, fcls = _ptr(filter_images), int(filter_class)
    else:
        fptr, fcls = None, -1
    sptr = None
    if stats is not None:
        assert stats.dtype == np.uint64 and stats.size >= 3
        sptr = _ptr(stats)
    rc = lib().rdf_oracle_eval_forest(_ptr(depth), n, w, h, _ptr(forest), T, D, C, fptr, fcls,
                                      _ptr(labels_out), int(labels_reduce), float(scale_factor),
                                      sptr, int(n_threads))
    if rc != 0:
        raise ValueError("rdf_oracle_eval_forest: bad arguments")
    return labels_out


def eval_tree(depth, tree, labels_out, n_threads=0):
    depth = _c(depth, np.uint16)
    tree = _c(tree, np.float32)
    assert tree.ndim == 2
    _, D, C = _forest_dims(tree[None])
    n, h, w = depth.shape
    assert labels_out.dtype == np.uint16 and labels_out.shape == (n, h, w) and labels_out.flags["C_CONTIGUOUS"]
    rc = lib().rdf_oracle_eval_tree(_ptr(depth), n, w, h, _ptr(tree), D, C, _ptr(labels_out), int(n_threads))
    if rc != 0:
        raise ValueError("rdf_oracle_eval_tree: bad arguments")
    return labels_out


def composite(label_images, conditions, out):
    """label_images: list of uint16 [H,W]; conditions int32 [K,2]; out uint16 [...,H,W] in place.

    Returns the number of pixels whose table walk was invalid (left untouched)."""
    imgs = [_c(a, np.uint16) for a in label_images]
    h, w = imgs[0].shape[-2:]
    cond = _c(conditions, np.int32).reshape(-1, 2)
    assert out.dtype == np.uint16 and out.flags["C_CONTIGUOUS"] and out.size == h * w
    table = (ctypes.c_void_p * len(imgs))(*[a.ctypes.data for a in imgs])
    bad = ctypes.c_int64(0)
    rc = lib().rdf_oracle_composite(table, len(imgs), w, h, _ptr(cond), cond.shape[0], _ptr(out),
                                    ctypes.byref(bad))
    if rc != 0:
        raise ValueError("rdf_oracle_composite: bad arguments")
    return int(bad.value)


def order_sensitive(depth, forest, labels_reduce=1, scale_factor=1.0):
    depth = _c(depth, np.uint16)
    forest = _c(forest, np.float32)
    n, h, w = depth.shape
    T, D, C = _forest_dims(forest)
    r = lib().rdf_oracle_order_sensitive(_ptr(depth), n, w, h, _ptr(forest), T, D, C, int(labels_reduce),
                                         float(scale_factor))
    if r < 0:
        raise ValueError("rdf_oracle_order_sensitive: bad arguments (T must be 1..4)")
    return int(r)


def distinct_nodes_per_level(depth, forest, labels_reduce=1, scale_factor=1.0, n_threads=0):
    """int64 [T][D]: how many different nodes of each level of each tree the batch's walks read."""
    depth = _c(depth, np.uint16)
    forest = _c(forest, np.float32)
    n, h, w = depth.shape
    T, D, C = _forest_dims(forest)
    vis = np.zeros((T, (1 << D) - 1), np.uint8)
    rc = lib().rdf_oracle_visit_map(_ptr(depth), n, w, h, _ptr(forest), T, D, C, int(labels_reduce), float(scale_factor),
                                    _ptr(vis), int(n_threads))
    if rc != 0:
        raise ValueError("rdf_oracle_visit_map: bad arguments")
    out = np.zeros((T, D), np.int64)
    for j in range(D):
        out[:, j] = vis[:, (1 << j) - 1:(1 << (j + 1)) - 1].sum(axis=1, dtype=np.int64)
    return out


def walk_lengths(depth, forest, labels_reduce=1, scale_factor=1.0, n_threads=0):
    """uint8 [N, H/r, W/r, T]: node records every walk reads (0: pixel not evaluated)."""
    depth = _c(depth, np.uint16)
    forest = _c(forest, np.float32)
    n, h, w = depth.shape
    T, D, C = _forest_dims(forest)
    out = np.zeros((n, h // labels_reduce, w // labels_reduce, T), np.uint8)
    rc = lib().rdf_oracle_walk_lengths(_ptr(depth), n, w, h, _ptr(forest), T, D, C, int(labels_reduce), float(scale_factor),
                                       _ptr(out), int(n_threads))
    if rc != 0:
        raise ValueError("rdf_oracle_walk_lengths: bad arguments")
    return out


def max_threads():
    return int(lib().rdf_oracle_max_threads())
