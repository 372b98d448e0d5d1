"""Constants the RDF path shares with its callers (counterpart of /root/reference/src/util.py:43)."""
import numpy as np

MAX_UINT16 = np.uint16(65535)  # "no pixel" marker in depth and label images


def make_grid(dims, block_dims):
    """Ceil-divide a 3-tuple of extents by a 3-tuple of block sizes."""
    assert len(dims) == 3 and len(block_dims) == 3
    return tuple((d + b - 1) // b for d, b in zip(dims, block_dims))
