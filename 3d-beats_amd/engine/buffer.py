"""`GpuBuffer` duck type of /root/reference/src/engine/buffer.py:10-39 without the OpenGL interop.

The reference's GpuBuffer is a GL buffer registered with CUDA; the RDF path only ever calls `.cu()`
on it (and reads `.shape` / `.dtype`).  Here `.cu()` returns the DeviceArray that owns the memory;
`.gl()` has no meaning on a headless MI355X box and raises.
"""
import numpy as np

from ..device import DeviceArray


class GpuBuffer:
    def __init__(self, shape, dtype, data_ptr=None, gl_buffer_flag=None):
        if isinstance(shape, (int, np.integer)):
            shape = (int(shape),)
        self.shape = tuple(shape)
        self.dtype = np.dtype(dtype)
        self._cu = DeviceArray(self.shape, self.dtype)
        if data_ptr is not None:
            self._cu.set(np.asarray(data_ptr, dtype=self.dtype).reshape(self.shape))

    def cu(self):
        return self._cu

    def gl(self):
        raise NotImplementedError("GL interop is outside the RDF inference path (no display on the GPU box)")
