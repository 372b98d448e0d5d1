"""Mean-shift mode finding on the composite label map, and fingertip heights.

Mirror of /root/reference/src/cuda/mean_shift.py: `MeanShift().run(num_rounds, labels, num_labels,
variances)` returns the per-class means as a host float64 array [num_labels, 2] = (x, y), exactly
like the reference (3d_bz.py:461-465).  All rounds run on the device in one C-ABI call
(`rdf_mean_shift`); the reference copies sums and means to the host and back every round.
"""
import numpy as np

from .. import _lib
from ..device import DeviceArray, device_ptr, get_runtime


class MeanShift:
    def __init__(self):
        self._rt = get_runtime()
        self._lib = self._rt.lib
        self.means = None       # DeviceArray float64 [num_labels, 2], as in the reference
        self._ws = None
        self._key = None

    def run_device(self, num_rounds, labels, num_labels, variances, out=None):
        """Same as run() but leaves the means on the device (DeviceArray float64 [num_labels, 2]); `out` = a float64
        device array of 2 * num_labels elements that receives them instead of self.means (HandPipeline's result
        buffer: no copy afterwards)."""
        dim_y, dim_x = labels.shape[-2:]
        key = (int(num_labels), int(num_rounds))
        if self._key != key:
            self.means = DeviceArray((num_labels, 2), np.float64)
            nbytes = int(self._lib.rdf_mean_shift_workspace_bytes(int(num_labels), int(num_rounds)))
            self._ws = DeviceArray((max(nbytes, 8),), np.uint8)
            self._key = key
        dst = self.means if out is None else out
        assert int(np.prod(dst.shape)) == 2 * int(num_labels) and np.dtype(dst.dtype) == np.float64
        rc = self._lib.rdf_mean_shift(device_ptr(labels), int(dim_x), int(dim_y), int(num_labels),
                                      device_ptr(variances), int(num_rounds), dst.ptr, self._ws.ptr,
                                      self._rt.stream())
        _lib.check(self._lib, rc, "rdf_mean_shift")
        dst.mark_dirty()
        return dst

    def run(self, num_rounds, labels, num_labels, variances):
        return self.run_device(num_rounds, labels, num_labels, variances).get()

    def run_device_with_heights(self, num_rounds, labels, num_labels, variances, class_ids, n_ids, depth_image,
                                labels_reduce, intrinsics, plane, means_out_ptr, heights_out_ptr):
        """run_device() and fingertip_heights() in ONE launch (rdf_mean_shift_heights): the same means and heights, bit for
        bit.  class_ids int32 [n_ids] and plane float32 [4,4] on the device; the two outputs are raw device-accessible
        pointers (float64 [num_labels, 2] and [n_ids]) -- HandPipeline hands in pinned host memory, so that the frame's
        result needs no copy."""
        dim_y, dim_x = labels.shape[-2:]
        ddy, ddx = depth_image.shape[-2:]
        fx, fy, ppx, ppy = (float(v) for v in intrinsics)
        rc = self._lib.rdf_mean_shift_heights(device_ptr(labels), int(dim_x), int(dim_y), int(num_labels),
                                              device_ptr(variances), int(num_rounds), int(means_out_ptr),
                                              device_ptr(class_ids), int(n_ids), device_ptr(depth_image), int(ddx), int(ddy),
                                              int(labels_reduce), fx, fy, ppx, ppy, device_ptr(plane), int(heights_out_ptr),
                                              self._rt.stream())
        _lib.check(self._lib, rc, "rdf_mean_shift_heights")


def fingertip_heights(means, class_ids, depth_image, labels_reduce, fx, fy, ppx, ppy, plane):
    """Device version of the per-fingertip height of 3d_bz.py:503-522.

    means: DeviceArray float64 [L,2] (MeanShift.run_device); class_ids: 1-based label ids (host sequence);
    depth_image: uint16 [H,W] on the device; plane: 4x4 float32 (host).  Returns host float64 [len(class_ids)],
    NaN where the reference resets the fingertip."""
    rt = get_runtime()
    lib = rt.lib
    ids = DeviceArray((len(class_ids),), np.int32).set(np.asarray(class_ids, dtype=np.int32))
    pl = DeviceArray((4, 4), np.float32).set(np.ascontiguousarray(plane, dtype=np.float32))
    out = DeviceArray((len(class_ids),), np.float64)
    dim_y, dim_x = depth_image.shape[-2:]
    rc = lib.rdf_fingertip_heights(device_ptr(means), int(means.shape[0]), ids.ptr, len(class_ids),
                                   device_ptr(depth_image), int(dim_x), int(dim_y), int(labels_reduce), float(fx),
                                   float(fy), float(ppx), float(ppy), pl.ptr, out.ptr, rt.stream())
    _lib.check(lib, rc, "rdf_fingertip_heights")
    return out.get()
