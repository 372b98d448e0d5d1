"""Counterpart of /root/reference/src/util.py for the RDF path (constants and grid helper only)."""
import numpy as np

MAX_UINT16 = np.uint16(65535)  # util.py:43 -- "no pixel" marker in depth and label images


def sizeof_fmt(num, suffix="B"):
    for unit in ["", "K", "M", "G", "T", "P", "E", "Z"]:
        if abs(num) < 1024.0:
            return "%3.1f %s%s" % (num, unit, suffix)
        num /= 1024.0
    return "%.1f%s%s" % (num, "Y", suffix)


def make_grid(dims, block_dims):
    assert len(dims) == 3 and len(block_dims) == 3
    return tuple(-(-a // b) for a, b in zip(dims, block_dims))
