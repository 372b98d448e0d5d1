"""Batch sharding across the GPUs of one node: one process per GPU, frames split contiguously,
forest replicated, ONE gather of label maps to rank 0.

The reference is single-process / single-GPU (src/engine/window.py:45-46); this module has no
counterpart there.  Frames are independent (tree_eval.cu:46-67 decodes purely by pixel index), so
the only communication is the gather, issued with torch.distributed (backend "nccl" = RCCL over
xGMI on the GPU box, "gloo" in the CPU tests).  RCCL runs gather as grouped send/recv, so each
peer's label maps ride that peer's own xGMI link into rank 0 instead of a ring.

The batch is evaluated in `n_chunks` pieces; the gather of chunk c is issued asynchronously right
after chunk c's kernel is enqueued, so it overlaps the evaluation of chunk c+1.

A second way to move the label maps, `PeerCopyGather` / `step_peer_copy`: rank `dst` exports its receive buffer once
(HIP IPC handle), every rank maps it and copies its shard into its slot with `hipMemcpyAsync` on a side stream after
each launch.  Copies between GPUs run on the copy engines over xGMI and occupy no CU, so they overlap the next
launch without taking compute units away from it -- which an RCCL kernel cannot do next to this forest kernel
(DESIGN.md section 6).  RCCL still does the control plane (handle broadcast, barriers).
"""
import ctypes

import numpy as np

from . import _lib
from .device import DeviceArray, get_runtime


def shard_range(n_frames, rank, world_size):
    """Contiguous split of the frame axis: rank g owns [g*n/W, (g+1)*n/W) (remainder to the low ranks)."""
    base, rem = divmod(int(n_frames), int(world_size))
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


class ShardedForestEvaluator:
    """Evaluates this rank's shard and gathers every rank's label maps on `dst`.

    All ranks must hold shards of the same shape (weak scaling: `frames_per_rank` each)."""

    def __init__(self, evaluator, forest, frames_per_rank, depth_dims, labels_reduce=1, scale_factor=1.,
                 n_chunks=4, dst=0, group=None):
        import torch.distributed as dist

        self.dist = dist
        self.ev = evaluator
        self.forest = forest
        self.group = group
        self.dst = dst
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.frames = int(frames_per_rank)
        self.h, self.w = int(depth_dims[0]), int(depth_dims[1])
        self.r = int(labels_reduce)
        self.s = scale_factor
        self.lh, self.lw = self.h // self.r, self.w // self.r
        n_chunks = max(1, min(int(n_chunks), self.frames))
        bounds = np.linspace(0, self.frames, n_chunks + 1).astype(int)
        self.chunks = [(int(a), int(b)) for a, b in zip(bounds[:-1], bounds[1:]) if b > a]
        # rank `dst` holds [world, frames, lh, lw]; everyone else only its own shard
        self._step_no = 0
        self._inflight = {}
        self.gathered = None
        if self.world > 1 and self.rank == dst:
            self.gathered = DeviceArray((self.world, self.frames, self.lh, self.lw), np.uint16)

    def _torch_view(self, arr, first_frame, n_frames):
        per = self.lh * self.lw * 2
        return arr.torch_bytes()[first_frame * per:(first_frame + n_frames) * per]

    def step(self, depth, labels, prefill=None):
        """depth: DeviceArray [frames,h,w]; labels: DeviceArray [frames,lh,lw] (this rank's output).

        Returns the list of outstanding gather handles (already waited on when this returns)."""
        assert tuple(depth.shape) == (self.frames, self.h, self.w)
        assert tuple(labels.shape) == (self.frames, self.lh, self.lw)
        if prefill is not None:
            labels.fill(prefill)
        works = []
        for a, b in self.chunks:
            self.ev.get_labels_forest(self.forest, depth[a:b], labels[a:b], labels_reduce=self.r,
                                      scale_factor=self.s)
            if self.world > 1:
                send = self._torch_view(labels, a, b - a)
                if self.rank == self.dst:
                    recv = [self._torch_view(self.gathered[g], a, b - a) for g in range(self.world)]
                else:
                    recv = None
                # stream-ordered after the kernel just enqueued; runs on the backend's own stream
                works.append(self.dist.gather(send, recv, dst=self.dst, group=self.group, async_op=True))
        for w in works:
            w.wait()
        return works

    # -- cross-step pipelining ----------------------------------------------------------------------
    def step_overlapped(self, depth, labels_ring, prefill=None):
        """One batch per call, ONE launch, and the gather of this step's label maps is left in flight: it
        overlaps the NEXT step's evaluation (labels_ring = two output buffers used alternately).  Call
        drain() before reading results or stopping the clock.  Steady state costs max(evaluate, gather)
        per step instead of their sum, without splitting the batch into smaller launches."""
        assert len(labels_ring) >= 2
        slot = self._step_no % len(labels_ring)
        labels = labels_ring[slot]
        assert tuple(depth.shape) == (self.frames, self.h, self.w)
        assert tuple(labels.shape) == (self.frames, self.lh, self.lw)
        pending = self._inflight.get(slot)
        if pending is not None:
            pending.wait()          # the evaluation below must not overwrite a buffer that is still being sent
            self._inflight[slot] = None
        if prefill is not None:
            labels.fill(prefill)
        self.ev.get_labels_forest(self.forest, depth, labels, labels_reduce=self.r, scale_factor=self.s)
        if self.world > 1:
            send = self._torch_view(labels, 0, self.frames)
            recv = None
            if self.rank == self.dst:
                recv = [self._torch_view(self.gathered[g], 0, self.frames) for g in range(self.world)]
            self._inflight[slot] = self.dist.gather(send, recv, dst=self.dst, group=self.group, async_op=True)
        self._step_no += 1
        return labels

    def drain(self):
        for slot, w in list(self._inflight.items()):
            if w is not None:
                w.wait()
                self._inflight[slot] = None

    def result(self):
        """Rank `dst`: DeviceArray [world*frames, lh, lw] of every rank's labels (world 1: None)."""
        if self.gathered is None:
            return None
        return self.gathered.reshape(self.world * self.frames, self.lh, self.lw)


class PeerCopyGather:
    """Receive buffer on rank `dst`, mapped into every process of the node.  Collective constructor.

    `ok` is False on every rank if any rank could not map the buffer (no IPC between the processes, e.g. devices hidden
    from each other): callers then fall back to the RCCL gather."""

    def __init__(self, world, rank, bytes_per_rank, dst=0, group=None, _fail_open_on_rank=None):
        import torch
        import torch.distributed as dist
        self._lib = get_runtime().lib
        self.world, self.rank, self.dst, self.bytes_per_rank = int(world), int(rank), int(dst), int(bytes_per_rank)
        self.base = None          # device pointer of the whole receive buffer as seen from THIS process
        self._owned, self._mapped = None, None
        handle = [None]
        mine_ok = 1
        if self.rank == self.dst:
            p = ctypes.c_void_p()
            rc = self._lib.rdf_device_malloc(ctypes.byref(p), self.world * self.bytes_per_rank)
            buf = ctypes.create_string_buffer(64)
            if rc == 0:
                self._owned = p.value
                rc = self._lib.rdf_ipc_export(p, buf)
            if rc == 0:
                handle[0] = bytes(buf.raw)
                self.base = self._owned
            else:
                mine_ok = 0
        dist.broadcast_object_list(handle, src=self.dst, group=group)
        if self.rank != self.dst:
            if handle[0] is None:
                mine_ok = 0
            else:
                p = ctypes.c_void_p()
                rc = self._lib.rdf_ipc_open(ctypes.create_string_buffer(handle[0], 64), ctypes.byref(p))
                if _fail_open_on_rank == self.rank:       # test hook: behave as if this rank could not map the buffer
                    if rc == 0:
                        self._lib.rdf_ipc_close(p)
                    rc = -1
                if rc == 0:
                    self._mapped = p.value
                    self.base = self._mapped
                else:
                    mine_ok = 0
        def all_ok(v):
            flag = torch.tensor([v], dtype=torch.int32)
            if dist.get_backend(group) == "nccl":
                flag = flag.cuda()
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
            return int(flag.item()) == 1

        self.ok = all_ok(mine_ok)
        if self.ok:
            # end-to-end check before anything relies on it: every rank copies a pattern into its slot, rank dst reads
            # all of them back
            n = min(256, self.bytes_per_rank)
            pattern = torch.full((n,), 17 + self.rank, dtype=torch.uint8, device="cuda")
            rc = self._lib.rdf_memcpy_device_async(ctypes.c_void_p(self.slot_ptr(self.rank)), ctypes.c_void_p(pattern.data_ptr()),
                                                   n, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
            torch.cuda.synchronize()
            copied = all_ok(1 if rc == 0 else 0)      # (also a barrier: every pattern has landed)
            seen = 1
            if copied and self.rank == self.dst:
                got = self.result_array().view(self.world, self.bytes_per_rank)[:, :n].cpu()
                want = (17 + torch.arange(self.world, dtype=torch.uint8)).view(-1, 1).expand(self.world, n)
                seen = 1 if bool((got == want).all()) else 0
            self.ok = copied and all_ok(seen)
        if not self.ok:
            self.close()

    def slot_ptr(self, rank, byte_offset=0):
        return self.base + rank * self.bytes_per_rank + byte_offset

    def result_array(self):
        """Rank `dst`: the receive buffer as a torch uint8 tensor (zero copy)."""
        import torch

        class _View:
            pass
        v = _View()
        v.__cuda_array_interface__ = {"shape": (self.world * self.bytes_per_rank,), "typestr": "|u1",
                                      "data": (self._owned, False), "version": 2}
        return torch.as_tensor(v, device="cuda")

    def close(self):
        if self._mapped is not None:
            self._lib.rdf_ipc_close(ctypes.c_void_p(self._mapped))
            self._mapped = None
        if self._owned is not None:
            self._lib.rdf_device_free(ctypes.c_void_p(self._owned))
            self._owned = None
        self.base = None


class PeerCopyForestEvaluator:
    """One launch per step; each rank copies its label maps into rank `dst`'s buffer on a side stream, so the copy
    of step s overlaps the launch of step s+1 (two label buffers used alternately, as in step_overlapped)."""

    def __init__(self, evaluator, forest, frames_per_rank, depth_dims, gather, labels_reduce=1, scale_factor=1.):
        import torch
        self.torch = torch
        self.ev, self.forest, self.gather = evaluator, forest, gather
        self.frames = int(frames_per_rank)
        self.h, self.w, self.r, self.s = int(depth_dims[0]), int(depth_dims[1]), int(labels_reduce), scale_factor
        self.lh, self.lw = self.h // self.r, self.w // self.r
        self.nbytes = self.frames * self.lh * self.lw * 2
        assert self.nbytes == gather.bytes_per_rank
        self._lib = get_runtime().lib
        self.copy_stream = torch.cuda.Stream()
        self._copied = {}
        self._step_no = 0

    def step(self, depth, labels_ring, prefill=None):
        torch = self.torch
        slot = self._step_no % len(labels_ring)
        labels = labels_ring[slot]
        assert tuple(labels.shape) == (self.frames, self.lh, self.lw)
        cur = torch.cuda.current_stream()
        if self._copied.get(slot) is not None:
            cur.wait_event(self._copied[slot])      # this buffer's previous contents have left
        if prefill is not None:
            labels.fill(prefill)
        self.ev.get_labels_forest(self.forest, depth, labels, labels_reduce=self.r, scale_factor=self.s)
        done = torch.cuda.Event()
        done.record(cur)
        self.copy_stream.wait_event(done)
        rc = self._lib.rdf_memcpy_device_async(ctypes.c_void_p(self.gather.slot_ptr(self.gather.rank)),
                                               ctypes.c_void_p(labels.ptr), self.nbytes,
                                               ctypes.c_void_p(self.copy_stream.cuda_stream))
        _lib.check(self._lib, rc, "rdf_memcpy_device_async")
        ev = torch.cuda.Event()
        ev.record(self.copy_stream)
        self._copied[slot] = ev
        self._step_no += 1
        return labels

    def drain(self):
        """This rank's copies have landed.  (Rank dst may read after a barrier that follows every rank's drain.)"""
        self.copy_stream.synchronize()

    def result(self):
        """Rank dst: DeviceArray-like torch view [world*frames, lh, lw] of uint16 as int16 bytes; else None."""
        if self.gather.rank != self.gather.dst:
            return None
        return self.gather.result_array().view(self.torch.int16).view(self.gather.world * self.frames, self.lh, self.lw)
