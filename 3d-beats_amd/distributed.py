"""Batch sharding across the GPUs of one node: one process per GPU, frames split contiguously,
forest replicated, ONE gather of label maps to rank 0.

The reference is single-process / single-GPU (src/engine/window.py:45-46); this module has no
counterpart there.  Frames are independent (tree_eval.cu:46-67 decodes purely by pixel index), so
the only communication is the gather, issued with torch.distributed (backend "nccl" = RCCL over
xGMI on the GPU box, "gloo" in the CPU tests).  RCCL runs gather as grouped send/recv, so each
peer's label maps ride that peer's own xGMI link into rank 0 instead of a ring.

The batch is evaluated in `n_chunks` pieces; the gather of chunk c is issued asynchronously right
after chunk c's kernel is enqueued, so it overlaps the evaluation of chunk c+1.
"""
import numpy as np

from .device import DeviceArray


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
