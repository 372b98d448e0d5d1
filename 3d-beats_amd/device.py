"""Device memory for the RDF path: a small GPUArray-like object and the runtime that backs it.

The reference passes PyCUDA `GPUArray`s (and `GpuBuffer` objects whose `.cu()` returns one,
/root/reference/src/engine/buffer.py:10-39) to the evaluator.  `DeviceArray` offers the members the
RDF path and its callers use: `.shape .dtype .size .nbytes .ptr .fill() .set() .get() .reshape()`,
first-axis indexing and `__cuda_array_interface__`.

Memory comes from a *runtime*: `HipRuntime` (torch-ROCm allocations + librdf_hip.so) is the only one
the package ships.  It raises when the library or a HIP device is missing -- there is no CPU
fallback.  Tests install a host-memory stand-in through `set_runtime()` to exercise host logic
without a GPU; that stand-in lives under tests/.
"""
import ctypes

import numpy as np

from . import _lib

_runtime = None


class HipRuntime:
    """torch-ROCm device memory + the HIP C ABI.  One per process; uses torch's current device/stream."""

    name = "hip"

    def __init__(self):
        self.lib = _lib.load()
        import torch

        if not torch.cuda.is_available():
            raise _lib.RdfError("no HIP device visible (torch.cuda.is_available() is False); "
                                "the RDF path has no CPU fallback")
        self.torch = torch

    # -- memory ------------------------------------------------------------------------------
    def alloc(self, nbytes):
        return self.torch.empty(max(int(nbytes), 1), dtype=self.torch.uint8, device="cuda")

    def alloc_host_mapped(self, nbytes):
        """Pinned host memory that kernels can write (hipHostMalloc memory is mapped into the device's address space at
        the same address on ROCm): (handle, device-accessible pointer, numpy uint8 view), or None if it cannot be had."""
        try:
            t = self.torch.empty(max(int(nbytes), 1), dtype=self.torch.uint8).pin_memory()
            return t, int(t.data_ptr()), t.numpy()
        except Exception:      # noqa: BLE001
            return None

    def ptr(self, handle):
        return int(handle.data_ptr())

    def _host_side(self, handle):
        """Host-mapped storage (alloc_host_mapped) is a pinned CPU tensor: copies to and from it are plain host memory
        traffic that no stream orders.  Kernels queued on the current stream may still be writing or reading it, so every
        host-side access waits for that stream first (the reference's `.cu().get()` right after a launch, run_live.py:131,
        then sees the finished labels, as a device buffer's synchronous copy does)."""
        if handle.device.type != "cpu":
            return False
        self.synchronize()
        return True

    def h2d(self, handle, offset, host_u8):
        if self._host_side(handle):
            handle.numpy()[offset:offset + host_u8.size] = host_u8
            return
        if host_u8.flags.writeable:
            handle[offset:offset + host_u8.size].copy_(self.torch.from_numpy(host_u8), non_blocking=False)
            return
        # a read-only source (a memory-mapped .npy): torch tensors cannot wrap it, so it goes through writable 64-MB pieces
        step = 1 << 26
        for a in range(0, host_u8.size, step):
            piece = np.array(host_u8[a:a + step])
            handle[offset + a:offset + a + piece.size].copy_(self.torch.from_numpy(piece), non_blocking=False)

    def d2h(self, handle, offset, nbytes):
        if self._host_side(handle):
            return handle.numpy()[offset:offset + nbytes].copy()
        return handle[offset:offset + nbytes].cpu().numpy()

    def as_torch(self, handle, offset, nbytes):
        return handle[offset:offset + nbytes]

    def d2d(self, dst, dst_off, src, src_off, nbytes):
        dst[dst_off:dst_off + nbytes].copy_(src[src_off:src_off + nbytes], non_blocking=True)

    def fill_bytes(self, handle, offset, nbytes, pattern_u8):
        """Fill [offset, offset+nbytes) with a repeating little-endian element pattern."""
        lib = self.lib
        if pattern_u8.size == 2 and nbytes % 2 == 0:
            v = int(pattern_u8.view(np.uint16)[0])
            _lib.check(lib, lib.rdf_fill_u16(self.ptr(handle) + offset, nbytes // 2, v, self.stream()),
                       "rdf_fill_u16")
            return
        t = self.torch
        self._host_side(handle)          # (a host write: nothing queued may still be using the memory)
        view = handle[offset:offset + nbytes]
        if len(set(pattern_u8.tolist())) == 1:
            view.fill_(int(pattern_u8[0]))
        else:
            reps = nbytes // pattern_u8.size
            view.copy_(t.from_numpy(np.tile(pattern_u8, reps)))

    # -- execution ---------------------------------------------------------------------------
    def stream(self):
        return int(self.torch.cuda.current_stream().cuda_stream)

    def synchronize(self):
        _lib.check(self.lib, self.lib.rdf_stream_synchronize(self.stream()), "rdf_stream_synchronize")


def get_runtime():
    """The process-wide runtime; created on first use.  Raises without a GPU or without the library."""
    global _runtime
    if _runtime is None:
        _runtime = HipRuntime()
    return _runtime


def set_runtime(rt):
    """Install a runtime (tests use this for a host-memory stand-in).  Returns the previous one."""
    global _runtime
    prev, _runtime = _runtime, rt
    return prev


class _Storage:
    __slots__ = ("handle", "nbytes", "version", "rt")

    def __init__(self, rt, nbytes):
        self.rt = rt
        self.handle = rt.alloc(nbytes)
        self.nbytes = nbytes
        self.version = 0


class _HostMappedStorage:
    """Pinned host memory that kernels can read and write (alloc_host_mapped) behind the same interface."""
    __slots__ = ("handle", "nbytes", "version", "rt", "host_view")

    def __init__(self, rt, nbytes):
        got = rt.alloc_host_mapped(nbytes)
        if got is None:
            raise _lib.RdfError("pinned, device-mapped host memory is not available")
        self.rt = rt
        self.handle, _, self.host_view = got
        self.nbytes = nbytes
        self.version = 0


def host_mapped_array(shape, dtype):
    """(DeviceArray, numpy array) over the SAME pinned host memory: kernels write it through the DeviceArray's pointer (the
    stores cross PCIe), the host reads the numpy array once the writing stream has been waited for."""
    dtype = np.dtype(dtype)
    shape = tuple(int(x) for x in (shape if not isinstance(shape, (int, np.integer)) else (shape,)))
    nbytes = int(np.prod(shape, dtype=np.int64)) * dtype.itemsize
    st = _HostMappedStorage(get_runtime(), max(nbytes, 1))
    arr = DeviceArray(shape, dtype, _storage=st)
    return arr, st.host_view[:nbytes].view(dtype).reshape(shape)


class DeviceArray:
    """C-contiguous n-d array in device memory (the GPUArray subset the RDF path needs)."""

    def __init__(self, shape, dtype, _storage=None, _offset=0):
        if isinstance(shape, (int, np.integer)):
            shape = (int(shape),)
        self.shape = tuple(int(s) for s in shape)
        self.dtype = np.dtype(dtype)
        self.size = int(np.prod(self.shape, dtype=np.int64)) if self.shape else 1
        self.nbytes = self.size * self.dtype.itemsize
        if _storage is None:
            _storage = _Storage(get_runtime(), self.nbytes)
        self._st = _storage
        self._off = int(_offset)
        assert self._off + self.nbytes <= max(self._st.nbytes, 1)

    # -- identity ----------------------------------------------------------------------------
    @property
    def ptr(self):
        return self._st.rt.ptr(self._st.handle) + self._off

    gpudata = ptr  # PyCUDA's name for the same thing

    @property
    def version(self):
        """Bumped by every mutation made through this API (any view of the same allocation)."""
        return self._st.version

    def mark_dirty(self):
        """Tell the cache logic that device code outside this API wrote the array."""
        self._st.version += 1

    @property
    def __cuda_array_interface__(self):
        return {"shape": self.shape, "typestr": self.dtype.str, "data": (self.ptr, False), "version": 2,
                "strides": None}

    def torch_bytes(self):
        """This array's bytes as a flat torch uint8 tensor sharing its memory (for torch.distributed)."""
        return self._st.rt.as_torch(self._st.handle, self._off, self.nbytes)

    # -- GPUArray-like surface ---------------------------------------------------------------
    def fill(self, value):
        pat = np.array([value]).astype(self.dtype).view(np.uint8)
        if self.nbytes:
            self._st.rt.fill_bytes(self._st.handle, self._off, self.nbytes, pat)
        self._st.version += 1
        return self

    def set(self, ary):
        a = np.ascontiguousarray(ary)
        assert a.dtype == self.dtype, f"dtype mismatch: {a.dtype} vs {self.dtype}"
        assert a.size == self.size, f"size mismatch: {a.shape} vs {self.shape}"
        if self.nbytes:
            self._st.rt.h2d(self._st.handle, self._off, a.reshape(-1).view(np.uint8))
        self._st.version += 1
        return self

    def copy_from(self, other):
        """Device-to-device copy of an array of the same size and dtype (stream-ordered)."""
        assert isinstance(other, DeviceArray) and other.dtype == self.dtype and other.size == self.size
        if self.nbytes:
            self._st.rt.d2d(self._st.handle, self._off, other._st.handle, other._off, self.nbytes)
        self._st.version += 1
        return self

    def get(self):
        if not self.nbytes:
            return np.empty(self.shape, self.dtype)
        raw = self._st.rt.d2h(self._st.handle, self._off, self.nbytes)
        return np.array(raw, copy=True).view(self.dtype).reshape(self.shape)

    def reshape(self, *shape):
        if len(shape) == 1 and isinstance(shape[0], (tuple, list)):
            shape = tuple(shape[0])
        out = DeviceArray(shape, self.dtype, self._st, self._off)
        assert out.size == self.size, f"cannot reshape {self.shape} to {shape}"
        return out

    def view(self, dtype):
        dtype = np.dtype(dtype)
        assert self.nbytes % dtype.itemsize == 0
        return DeviceArray((self.nbytes // dtype.itemsize,), dtype, self._st, self._off)

    def __getitem__(self, idx):
        """First-axis int or contiguous slice (e.g. forest_cu[i].set(tree), decision_tree.py:591)."""
        assert len(self.shape) >= 1
        inner = int(np.prod(self.shape[1:], dtype=np.int64)) * self.dtype.itemsize
        if isinstance(idx, (int, np.integer)):
            i = int(idx) + (self.shape[0] if idx < 0 else 0)
            assert 0 <= i < self.shape[0]
            return DeviceArray(self.shape[1:], self.dtype, self._st, self._off + i * inner)
        if isinstance(idx, slice):
            start, stop, step = idx.indices(self.shape[0])
            assert step == 1, "only contiguous first-axis slices"
            n = max(stop - start, 0)
            return DeviceArray((n,) + self.shape[1:], self.dtype, self._st, self._off + start * inner)
        raise TypeError("DeviceArray supports first-axis int/slice indexing only")

    def __len__(self):
        return self.shape[0]

    def __repr__(self):
        return f"DeviceArray(shape={self.shape}, dtype={self.dtype}, ptr=0x{self.ptr:x})"


def to_device(ary):
    a = np.ascontiguousarray(ary)
    return DeviceArray(a.shape, a.dtype).set(a)


def zeros(shape, dtype):
    return DeviceArray(shape, dtype).fill(0)


def device_ptr(obj):
    """Raw device address of a DeviceArray / GpuBuffer / torch tensor / __cuda_array_interface__ object."""
    if obj is None:
        return None
    if isinstance(obj, DeviceArray):
        return obj.ptr
    if hasattr(obj, "cu") and callable(obj.cu):
        return device_ptr(obj.cu())
    if hasattr(obj, "data_ptr"):
        return int(obj.data_ptr())
    if hasattr(obj, "__cuda_array_interface__"):
        return int(obj.__cuda_array_interface__["data"][0])
    if hasattr(obj, "ptr"):
        return int(obj.ptr)
    raise TypeError(f"cannot take a device pointer from {type(obj)!r}")


def ptr_arg(obj):
    p = device_ptr(obj)
    return ctypes.c_void_p(p) if p is not None else None
