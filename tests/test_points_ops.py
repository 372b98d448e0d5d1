"""the element-wise kernels that define the forest's input convention and colour its output.
CPU: hand-derived answers for the numpy restatement.  GPU: byte-exact against the restatement."""
import importlib

import numpy as np
import pytest

from oracle import points_ops_numpy as po_np


def test_known_answers_numpy():
    d = np.array([[0, 5, 65535], [7, 0, 1]], np.uint16)
    assert po_np.convert_0s_to_maxuint(d.copy()).tolist() == [[65535, 5, 65535], [7, 65535, 1]]
    pts = np.ones((2, 3, 4), np.float32)
    pts[0, 1, 3] = 0.0                      # filtered point: w == 0 (points_ops.cu:160-162)
    assert po_np.setup_depth_image_for_forest(pts, d.copy()).tolist() == [[65535, 65535, 65535], [7, 65535, 1]]
    # stencil at mip level 1 on a 5x3 image: groups are 2x1 (integer division), the odd last row/column reads
    # out of bounds = group 0
    g = np.array([[1, 2]], np.uint16)
    din = np.arange(15, dtype=np.uint16).reshape(3, 5) + 100
    out = po_np.stencil_depth_image_by_group(5, 3, 1, 2, g, din, np.zeros((3, 5), np.uint16))
    assert out.tolist() == [[0, 0, 102, 103, 0], [0, 0, 107, 108, 0], [0, 0, 0, 0, 0]]
    out0 = po_np.stencil_depth_image_by_group(5, 3, 1, 0, g, din, np.zeros((3, 5), np.uint16))
    assert out0.tolist() == [[0, 0, 0, 0, 104], [0, 0, 0, 0, 109], [110, 111, 112, 113, 114]]
    assert po_np.flip_x(np.array([[1, 2, 3]], np.uint16), np.zeros((1, 3), np.uint16)).tolist() == [[3, 2, 1]]
    cols = np.array([[1, 2, 3, 4], [5, 6, 7, 8]], np.uint8)
    img = po_np.make_rgba_from_labels(np.array([[0, 1, 2, 3, 65535]], np.uint16), cols, np.full((1, 5, 4), 9, np.uint8))
    assert img.tolist() == [[[9] * 4, [1, 2, 3, 4], [5, 6, 7, 8], [9] * 4, [9] * 4]]


@pytest.mark.gpu
def test_pointsops_byte_exact_on_gpu(rdf, gpu_runtime):
    pmod = importlib.import_module("3d-beats_amd.cuda.points_ops")
    ops = pmod.PointsOps()
    rng = np.random.default_rng(21)
    for (h, w) in [(480, 848), (37, 53), (1, 1), (240, 424)]:
        depth = rdf.synth.live_frame(5, h, w) if h > 8 else np.array([[0]], np.uint16)
        n = h * w
        # convert_0s_to_maxuint, also on unaligned sub-ranges
        for off in (0, 1, 3):
            if off >= n:
                continue
            dev = rdf.to_device(depth.reshape(-1))
            ops.convert_0s_to_maxuint(np.int32(n - off), dev.view(np.uint16)[off:] if off else dev)
            want = depth.reshape(-1).copy()
            po_np.convert_0s_to_maxuint(want[off:])
            assert np.array_equal(dev.get(), want), (h, w, off)
        # setup_depth_image_for_forest
        pts = rng.standard_normal((h, w, 4)).astype(np.float32)
        pts[..., 3] = (rng.random((h, w)) > 0.3).astype(np.float32)
        dev = rdf.to_device(depth)
        ops.setup_depth_image_for_forest(np.int32(n), rdf.to_device(pts), dev)
        assert np.array_equal(dev.get(), po_np.setup_depth_image_for_forest(pts, depth.copy()))
        # stencil by group at mip levels 0..3
        for level in (0, 1, 3):
            f = 1 << level
            gw, gh = max(w // f, 1), max(h // f, 1)
            groups = rng.integers(0, 3, size=(gh, gw)).astype(np.uint16)
            for group in (1, 0):
                out = rdf.to_device(np.full((h, w), 7, np.uint16))
                ops.stencil_depth_image_by_group(np.array([w, h], np.int32), np.int32(level), np.int32(group),
                                                 rdf.to_device(groups), rdf.to_device(depth), out)
                want = po_np.stencil_depth_image_by_group(w, h, level, group, groups if w // f and h // f else groups[:0],
                                                          depth, np.full((h, w), 7, np.uint16))
                assert np.array_equal(out.get(), want), (h, w, level, group)
        # flip_x
        out = rdf.DeviceArray((h, w), np.uint16)
        ops.flip_x(np.array([w, h], np.int32), rdf.to_device(depth), out)
        assert np.array_equal(out.get(), depth[:, ::-1])
        # make_rgba_from_labels
        labels = rng.integers(0, 9, size=(h, w)).astype(np.uint16)
        labels[rng.random((h, w)) < 0.2] = 65535
        cols = rng.integers(0, 256, size=(6, 4)).astype(np.uint8)
        img = rdf.to_device(np.full((h, w, 4), 3, np.uint8))
        ops.make_rgba_from_labels(np.uint32(w), np.uint32(h), np.uint32(6), rdf.to_device(labels), rdf.to_device(cols), img)
        assert np.array_equal(img.get(), po_np.make_rgba_from_labels(labels, cols, np.full((h, w, 4), 3, np.uint8)))


@pytest.mark.gpu
def test_prepare_hand_depth_equals_the_chain_it_replaces(rdf, gpu_runtime):
    """rdf_prepare_hand_depth == fill(0) -> stencil_depth_image_by_group -> flip_x (or copy) -> convert_0s_to_maxuint
    (3d_bz.py:396-420), byte for byte: vector and scalar widths, every mip level, both orientations, in place."""
    pmod = importlib.import_module("3d-beats_amd.cuda.points_ops")
    ops = pmod.PointsOps()
    rng = np.random.default_rng(33)
    for (h, w) in [(480, 848), (37, 53), (1, 1), (240, 424), (16, 8), (9, 24)]:
        depth = rdf.synth.live_frame(9, h, w) if h > 8 else rng.integers(0, 3, size=(h, w)).astype(np.uint16)
        depth[rng.random((h, w)) < 0.1] = 0
        for level in (0, 1, 3):
            f = 1 << level
            gw, gh = max(w // f, 1), max(h // f, 1)
            groups = rng.integers(0, 3, size=(gh, gw)).astype(np.uint16)
            g_host = groups if w // f and h // f else groups[:0]
            for group in (1, 2):
                for flip in (False, True):
                    want = po_np.stencil_depth_image_by_group(w, h, level, group, g_host, depth, np.zeros((h, w), np.uint16))
                    want = want[:, ::-1].copy() if flip else want
                    po_np.convert_0s_to_maxuint(want)
                    out = rdf.to_device(np.full((h, w), 7, np.uint16))
                    ops.prepare_hand_depth(np.array([w, h], np.int32), level, group, rdf.to_device(groups), rdf.to_device(depth),
                                           out, flip)
                    assert np.array_equal(out.get(), want), (h, w, level, group, flip)
            inplace = rdf.to_device(depth)
            ops.prepare_hand_depth(np.array([w, h], np.int32), level, 1, rdf.to_device(groups), inplace, inplace, False)
            want = po_np.convert_0s_to_maxuint(po_np.stencil_depth_image_by_group(w, h, level, 1, g_host, depth,
                                                                                  np.zeros((h, w), np.uint16)))
            assert np.array_equal(inplace.get(), want)
    d = rdf.to_device(np.ones((4, 8), np.uint16))
    rc = gpu_runtime.lib.rdf_prepare_hand_depth(8, 4, 0, 1, d.ptr, d.ptr, d.ptr, 1, gpu_runtime.stream())
    assert rc == -1      # a flip in place is refused
