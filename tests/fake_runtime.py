"""Host-memory stand-in for the package's HipRuntime -- TEST DOUBLE, lives under tests/ only.

It lets the host logic of 3d-beats_amd/decision_tree.py (shape checks, layer wiring, pack caching,
batch chunking, the distributed driver) run in a container without a GPU: "device" memory is numpy
memory and the C-ABI entry points are answered by the CPU oracle on those host pointers.  Nothing
in the shipped package imports this file.
"""
import ctypes

import numpy as np

from oracle import rdf_oracle


class _OracleLib:
    """Same entry-point names and argument order as include/rdf_hip.h, served by oracle/rdf_oracle.c."""

    def __init__(self):
        self._o = rdf_oracle.lib()
        self.calls = []
        self.trace = []          # (entry point, every argument as passed): tests/test_callconv.py reads the buffer wiring from it
        self._packed_scale = {}

    def rdf_eval_forest(self, depth, n_img, dim_x, dim_y, forest, T, D, C, filt, fcls, out, r, s, stream):
        self.trace.append(("rdf_eval_forest", (depth, n_img, dim_x, dim_y, forest, T, D, C, filt, fcls, out, r, s, stream,)))
        self.calls.append(("rdf_eval_forest", n_img, dim_x, dim_y, T, D, C, fcls, r, float(s)))
        rc = self._o.rdf_oracle_eval_forest(depth, n_img, dim_x, dim_y, forest, T, D, C, filt, fcls, out, r, s, None, 0)
        return 0 if rc == 0 else -1

    def rdf_eval_forest_stats(self, depth, n_img, dim_x, dim_y, forest, T, D, C, filt, fcls, out, r, s, stats, stream):
        self.calls.append(("rdf_eval_forest_stats", n_img, dim_x, dim_y, T, D, C, fcls, r, float(s)))
        rc = self._o.rdf_oracle_eval_forest(depth, n_img, dim_x, dim_y, forest, T, D, C, filt, fcls, out, r, s, stats, 0)
        return 0 if rc == 0 else -1

    def rdf_eval_tree(self, depth, n_img, dim_x, dim_y, tree, D, C, out, stream):
        self.trace.append(("rdf_eval_tree", (depth, n_img, dim_x, dim_y, tree, D, C, out, stream,)))
        self.calls.append(("rdf_eval_tree", n_img, dim_x, dim_y, D, C))
        rc = self._o.rdf_oracle_eval_tree(depth, n_img, dim_x, dim_y, tree, D, C, out, 0)
        return 0 if rc == 0 else -1

    def rdf_forest_packed_bytes(self, T, D, C):
        return (T << D) * (48 + 8 * ((C + 3) & ~3)) + ((T << (D - 1)) * 64 + 64 if 1 <= C <= 4 and D >= 2 and T >= 1 else 0)

    def rdf_forest_pack(self, forest, T, D, C, s, packed, stream):
        self.calls.append(("rdf_forest_pack", T, D, C, float(s)))
        self._packed_scale[int(packed)] = float(s)
        return 0

    def rdf_eval_forest_packed(self, depth, n_img, dim_x, dim_y, packed, forest, T, D, C, filt, fcls, out, r, stream):
        self.trace.append(("rdf_eval_forest_packed", (depth, n_img, dim_x, dim_y, packed, forest, T, D, C, filt, fcls, out, r, stream,)))
        s = self._packed_scale[int(packed)]
        self.calls.append(("rdf_eval_forest_packed", n_img, dim_x, dim_y, T, D, C, fcls, r, s))
        rc = self._o.rdf_oracle_eval_forest(depth, n_img, dim_x, dim_y, forest, T, D, C, filt, fcls, out, r, s, None, 0)
        return 0 if rc == 0 else -1

    def rdf_eval_forest_packed_filled(self, depth, n_img, dim_x, dim_y, packed, forest, T, D, C, filt, fcls, out, r, stream):
        self.rdf_fill_u16(out, int(n_img) * (int(dim_y) // int(r)) * (int(dim_x) // int(r)), 65535, stream)
        return self.rdf_eval_forest_packed(depth, n_img, dim_x, dim_y, packed, forest, T, D, C, filt, fcls, out, r, stream)

    def rdf_composite(self, images, n_images, dim_x, dim_y, cond, n_cond, out, bad, stream):
        self.trace.append(("rdf_composite", (images, n_images, dim_x, dim_y, cond, n_cond, out, bad, stream,)))
        self.calls.append(("rdf_composite", n_images, dim_x, dim_y, n_cond))
        nb = ctypes.c_int64(0)
        rc = self._o.rdf_oracle_composite(images, n_images, dim_x, dim_y, cond, n_cond, out, ctypes.byref(nb))
        if bad:
            ctypes.cast(bad, ctypes.POINTER(ctypes.c_int32))[0] += int(nb.value)
        return 0 if rc == 0 else -1

    def rdf_layered_run(self, depth, dim_x, dim_y, n_layers, packed, forests, n_trees, max_depth, n_classes,
                        filter_layer, filter_class, layer_labels, table_dev, cond, n_cond, out, bad, r, s, stream):
        self.trace.append(("rdf_layered_run", (depth, dim_x, dim_y, n_layers, packed, forests, n_trees, max_depth, n_classes, filter_layer, filter_class, layer_labels, table_dev, cond, n_cond, out, bad, r, s, stream,)))
        self.calls.append(("rdf_layered_run", dim_x, dim_y, n_layers, r, float(s)))
        lw, lh = dim_x // r, dim_y // r
        self.rdf_fill_u16(out, lw * lh, 65535, 0)
        for i in range(n_layers):
            self.rdf_fill_u16(layer_labels[i], lw * lh, 65535, 0)
        self.calls = [c for c in self.calls if c[0] != "rdf_fill_u16" or c is None]
        for i in range(n_layers):
            fl = filter_layer[i]
            rc = self._o.rdf_oracle_eval_forest(depth, 1, dim_x, dim_y, forests[i], n_trees[i], max_depth[i],
                                                n_classes[i], layer_labels[fl] if fl >= 0 else None,
                                                filter_class[i] if fl >= 0 else -1, layer_labels[i], r, s, None, 0)
            if rc != 0:
                return -1
        return self.rdf_composite(table_dev, n_layers, lw, lh, cond, n_cond, out, bad, stream)

    def rdf_fill_u16(self, dst, n, value, stream):
        self.trace.append(("rdf_fill_u16", (dst, n, value, stream,)))
        self.calls.append(("rdf_fill_u16", int(n), int(value)))
        arr = (ctypes.c_uint16 * int(n)).from_address(int(dst))
        np.frombuffer(arr, dtype=np.uint16)[:] = value
        return 0

    def rdf_convert_0s_to_maxuint(self, depth, n, stream):       # points_ops.cu:117-127
        self.calls.append(("rdf_convert_0s_to_maxuint", int(n)))
        a = np.frombuffer((ctypes.c_uint16 * int(n)).from_address(int(depth)), dtype=np.uint16)
        a[a == 0] = 65535
        return 0

    def rdf_stream_synchronize(self, stream):
        return 0

    def rdf_error_string(self, code):
        return b"fake runtime error"


class HostRuntime:
    name = "host-test-double"

    def __init__(self):
        self.lib = _OracleLib()

    def alloc(self, nbytes):
        return np.zeros(max(int(nbytes), 1), dtype=np.uint8)

    def ptr(self, handle):
        return int(handle.ctypes.data)

    def h2d(self, handle, offset, host_u8):
        handle[offset:offset + host_u8.size] = host_u8

    def d2h(self, handle, offset, nbytes):
        return handle[offset:offset + nbytes].copy()

    def fill_bytes(self, handle, offset, nbytes, pattern_u8):
        if pattern_u8.size == 2 and nbytes % 2 == 0:
            self.lib.rdf_fill_u16(self.ptr(handle) + offset, nbytes // 2, int(pattern_u8.view(np.uint16)[0]), 0)
        else:
            handle[offset:offset + nbytes] = np.tile(pattern_u8, nbytes // pattern_u8.size)

    def d2d(self, dst, dst_off, src, src_off, nbytes):
        dst[dst_off:dst_off + nbytes] = src[src_off:src_off + nbytes]

    def as_torch(self, handle, offset, nbytes):
        import torch
        return torch.from_numpy(handle)[offset:offset + nbytes]

    def stream(self):
        return 0

    def synchronize(self):
        pass
