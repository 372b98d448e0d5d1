"""numpy restatement of the five element-wise kernels of SURVEY 8f-2.  TEST INFRASTRUCTURE ONLY.
Follows /root/reference/src/cuda/points_ops.cu:117-127, 149-165, 440-463, 466-483, 258-281.
Integer / byte exact; parity unpinned by reference fixtures (the reference has none)."""
import numpy as np


def convert_0s_to_maxuint(depth):
    depth[depth == 0] = 65535
    return depth


def setup_depth_image_for_forest(pts, depth):
    flat, p = depth.reshape(-1), pts.reshape(-1, 4)
    flat[(flat == 0) | (p[:, 3] == 0)] = 65535
    return depth


def stencil_depth_image_by_group(dim_x, dim_y, level, group, g_in, d_in, d_out):
    f = 1 << level
    gw, gh = dim_x // f, dim_y // f
    g = np.asarray(g_in).reshape(-1)[: gw * gh].reshape(gh, gw)
    yy, xx = np.mgrid[0:dim_y, 0:dim_x]
    gy, gx = yy // f, xx // f
    inb = (gy < gh) & (gx < gw)
    gv = np.zeros((dim_y, dim_x), dtype=np.int64)       # Array2d::get returns its default (0) out of bounds
    gv[inb] = g[gy[inb], gx[inb]]
    m = gv == group
    d_out.reshape(dim_y, dim_x)[m] = d_in.reshape(dim_y, dim_x)[m]
    return d_out


def flip_x(img_in, img_out):
    img_out[...] = img_in[..., ::-1]
    return img_out


def make_rgba_from_labels(labels, colors, image):
    lab = labels.reshape(labels.shape[-2], labels.shape[-1]).astype(np.int64)
    m = (lab != 0) & (lab != 65535) & (lab <= colors.shape[0])
    image.reshape(lab.shape + (4,))[m] = colors[lab[m] - 1]
    return image
