"""The element-wise kernels either side of the forest, callable the way the reference's
apps call their PyCUDA counterparts (/root/reference/src/cuda/points_ops.py:16-44;
src/3d_bz.py:396-456; src/run_live_layered.py:117-122): positional arguments as there, `grid=` and
`block=` accepted and ignored (launch geometry belongs to the library).  Only the five kernels that
touch the forest's input and output live here; deprojection, plane fitting, filtering and mesh
generation stay out of scope."""
import numpy as np

from .. import _lib
from ..device import device_ptr, get_runtime


class PointsOps:
    def __init__(self):
        self._rt = get_runtime()
        self._lib = self._rt.lib

    def _ok(self, rc, name, *touched):
        _lib.check(self._lib, rc, name)
        for t in touched:
            if hasattr(t, "mark_dirty"):
                t.mark_dirty()

    def convert_0s_to_maxuint(self, num_pixels, depth, grid=None, block=None):
        self._ok(self._lib.rdf_convert_0s_to_maxuint(device_ptr(depth), int(num_pixels), self._rt.stream()),
                 "rdf_convert_0s_to_maxuint", depth)

    def setup_depth_image_for_forest(self, num_pixels, pts, depth, grid=None, block=None):
        self._ok(self._lib.rdf_setup_depth_image_for_forest(device_ptr(pts), device_ptr(depth), int(num_pixels),
                                                            self._rt.stream()),
                 "rdf_setup_depth_image_for_forest", depth)

    def stencil_depth_image_by_group(self, img_dim, mipmap_level, group, g_in, d_in, d_out, grid=None, block=None):
        dim_x, dim_y = (int(v) for v in np.asarray(img_dim).reshape(-1)[:2])
        self._ok(self._lib.rdf_stencil_depth_image_by_group(dim_x, dim_y, int(mipmap_level), int(group), device_ptr(g_in),
                                                            device_ptr(d_in), device_ptr(d_out), self._rt.stream()),
                 "rdf_stencil_depth_image_by_group", d_out)

    def prepare_hand_depth(self, img_dim, mipmap_level, group, g_in, d_in, d_out, flip_x):
        """fill(0) + stencil_depth_image_by_group + flip_x (or copy) + convert_0s_to_maxuint of 3d_bz.py:396-420 as one
        pass: d_out = the frame as the forest wants it for this hand.  No reference counterpart as a single kernel."""
        dim_x, dim_y = (int(v) for v in np.asarray(img_dim).reshape(-1)[:2])
        self._ok(self._lib.rdf_prepare_hand_depth(dim_x, dim_y, int(mipmap_level), int(group), device_ptr(g_in),
                                                  device_ptr(d_in), device_ptr(d_out), 1 if flip_x else 0, self._rt.stream()),
                 "rdf_prepare_hand_depth", d_out)

    def flip_x(self, img_dim, img_in, img_out, grid=None, block=None):
        dim_x, dim_y = (int(v) for v in np.asarray(img_dim).reshape(-1)[:2])
        self._ok(self._lib.rdf_flip_x(dim_x, dim_y, device_ptr(img_in), device_ptr(img_out), self._rt.stream()),
                 "rdf_flip_x", img_out)

    def make_rgba_from_labels(self, dim_x, dim_y, num_colors, labels, colors, color_image, grid=None, block=None):
        self._ok(self._lib.rdf_make_rgba_from_labels(int(dim_x), int(dim_y), int(num_colors), device_ptr(labels),
                                                     device_ptr(colors), device_ptr(color_image), self._rt.stream()),
                 "rdf_make_rgba_from_labels", color_image)
