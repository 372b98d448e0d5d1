"""CPU restatement of the reference's mean-shift mode finding and fingertip height.

TEST INFRASTRUCTURE ONLY.  Follows /root/reference/src/cuda/mean_shift.cu:3-48 (per-pixel terms),
src/cuda/mean_shift.py:35-59 (host loop: zero sums, accumulate, means += sums[:, :2] / sums[:, 2]) and
src/3d_bz.py:503-522 (height).  The reference accumulates with fp64 atomicAdd in scheduler order, so its
own results vary in the last bits from run to run: parity for this row is a tolerance (1e-9 pixels),
not bit-exactness.  `rs2_deproject_pixel_to_point` comes from librealsense2 (pyrealsense2, unpinned
in the reference's requirements.txt, absent here); its distortion-free branch is restated from the
published rsutil.h: x = (px - ppx)/fx, y = (py - ppy)/fy in fp32, point = depth * (x, y, 1).
"""
import numpy as np


def mean_shift(labels, num_labels, variances, num_rounds):
    lab = np.asarray(labels).reshape(labels.shape[-2], labels.shape[-1]).astype(np.int64)
    h, w = lab.shape
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    var = np.asarray(variances, dtype=np.float32)
    means = np.zeros((num_labels, 2), dtype=np.float64)
    for rnd in range(num_rounds):
        sums = np.zeros((num_labels, 3), dtype=np.float64)
        for c in range(num_labels):
            m = lab == (c + 1)
            if not m.any():
                continue
            x, y = xx[m], yy[m]
            if rnd == 0:
                sums[c] = [x.sum(), y.sum(), float(m.sum())]
            else:
                dx, dy = x - means[c, 0], y - means[c, 1]
                v2 = np.float64(np.float32(var[c] * var[c]))       # float product (mean_shift.cu:42)
                with np.errstate(invalid="ignore"):
                    p = np.exp(-((dx * dx) + (dy * dy)) / (2 * v2))
                sums[c] = [(dx * p).sum(), (dy * p).sum(), p.sum()]
        with np.errstate(invalid="ignore", divide="ignore"):
            means += sums[:, 0:2] / sums[:, 2].reshape((num_labels, 1))
    return means


def fingertip_heights(means, class_ids, depth, labels_reduce, fx, fy, ppx, ppy, plane):
    depth = np.asarray(depth).reshape(depth.shape[-2], depth.shape[-1])
    h, w = depth.shape
    plane = np.asarray(plane, dtype=np.float32)
    out = np.full(len(class_ids), np.nan, dtype=np.float64)
    for i, c in enumerate(class_ids):
        m = means[c - 1]
        if np.isnan(m).any():
            continue
        px, py = int(m[0]) * labels_reduce, int(m[1]) * labels_reduce
        if px < 0 or py < 0 or px >= w or py >= h:
            continue
        z = np.float32(depth[py, px])
        x = (np.float32(px) - np.float32(ppx)) / np.float32(fx)
        y = (np.float32(py) - np.float32(ppy)) / np.float32(fy)
        pt = np.array([float(z * x), float(z * y), float(z), 1.0], dtype=np.float64)
        out[i] = -(plane @ pt)[2]
    return out
