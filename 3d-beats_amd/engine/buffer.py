"""`GpuBuffer` duck type of /root/reference/src/engine/buffer.py:10-39 without the OpenGL interop.

The reference's GpuBuffer is a GL buffer registered with CUDA; the RDF path only ever calls `.cu()`
on it (and reads `.shape` / `.dtype`).  Here `.cu()` returns the DeviceArray that owns the memory;
`.gl()` has no meaning on a headless MI355X box and raises.
"""
import numpy as np

from ..device import DeviceArray, host_mapped_array


class GpuBuffer:
    def __init__(self, shape, dtype, data_ptr=None, gl_buffer_flag=None, host_mapped=False):
        """host_mapped (not in the reference): the buffer lives in pinned host memory mapped into the device; a kernel that
        writes it stores across PCIe and `.host` is the same memory as a numpy array -- a result buffer that needs no read
        back (valid once the writing stream has been waited for)."""
        if isinstance(shape, (int, np.integer)):
            shape = (int(shape),)
        self.shape = tuple(shape)
        self.dtype = np.dtype(dtype)
        self.host = None
        if host_mapped:
            self._cu, self.host = host_mapped_array(self.shape, self.dtype)
        else:
            self._cu = DeviceArray(self.shape, self.dtype)
        if data_ptr is not None:
            self._cu.set(np.asarray(data_ptr, dtype=self.dtype).reshape(self.shape))

    def cu(self):
        return self._cu

    def gl(self):
        raise NotImplementedError("GL interop is outside the RDF inference path (no display on the GPU box)")
