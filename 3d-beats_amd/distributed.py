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
from .device import DeviceArray, _Storage, get_runtime


def shard_range(n_frames, rank, world_size):
    """Contiguous split of the frame axis: rank g owns [g*n/W, (g+1)*n/W) (remainder to the low ranks)."""
    base, rem = divmod(int(n_frames), int(world_size))
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def replicate_forest(forest, src=0, group=None, packed_scales=()):
    """The forest is REPLICATED on every GPU: one broadcast of rank `src`'s `forest_cu` (240 MiB for T4/D20, 1.9 GiB
    for T8/D22) outside any timed region, instead of every rank reading the .npy.  Every rank passes a DecisionForest of the same
    trees / depth / classes; the other ranks' contents are overwritten.  With `packed_scales`, rank `src`'s PACKED tables for those
    scale factors go over as well -- deep-level choice included, it lives in the table -- so rank `src` packs and tunes once
    (`forest.packed(s)`, `forest.tune(sample, scale_factor=s)`) and no other rank packs or tunes: every rank walks the same table.
    Collective; returns `forest`."""
    import torch.distributed as dist
    from .decision_tree import _touch
    rank = dist.get_rank(group)
    dist.broadcast(forest.forest_cu.torch_bytes(), src=src, group=group)
    if rank != src:
        _touch(forest.forest_cu)                    # (written outside the array API: the packed-table cache must notice)
    if packed_scales:
        from .device import DeviceArray, get_runtime
        lib = get_runtime().lib
        nbytes = int(lib.rdf_forest_packed_bytes(int(forest.num_trees), int(forest.max_depth), int(forest.num_classes)))
        for s_ in packed_scales:
            s = float(np.float32(s_))
            if nbytes == 0:
                continue
            if rank == src:
                buf = forest.packed(s)
            else:
                hit = forest._packed.get(s)
                buf = hit[1] if (hit is not None and hit[1].nbytes == nbytes) else DeviceArray((nbytes,), np.uint8)
                forest._forget(buf)                 # whatever the library knew about this address
                forest._packed.pop(s, None)
            dist.broadcast(buf.torch_bytes(), src=src, group=group)
            if rank != src:
                # the library's first look at the received bytes, now (magic, shape, generation; the scale against `s`): a bad
                # table raises ValueError here instead of evaluating with another scale later (DecisionForest._verify_table)
                forest._verify_table(buf, s, f"replicate_forest (rank {rank})")
                forest._packed[s] = ((id(forest.forest_cu), forest.forest_cu.version), buf)
                forest.__dict__.setdefault("_tuned", {}).pop(s, None)
    return forest


class ShardedForestEvaluator:
    """Evaluates this rank's shard and gathers every rank's label maps on `dst`.

    All ranks must hold shards of the same shape (weak scaling: `frames_per_rank` each)."""

    def __init__(self, evaluator, forest, frames_per_rank, depth_dims, labels_reduce=1, scale_factor=1.,
                 n_chunks=4, dst=0, group=None, helper_cus=0):
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
        # step_overlapped on a CU-masked compute stream (rdf_stream_create_with_reserved_cus): the `helper_cus` compute units the
        # stream leaves to RCCL's kernel are idle once the gather has finished -- about half of a step.  With helper_cus > 0 every
        # step is a SPLIT launch (DecisionTreeEvaluator.get_labels_forest_split): the main launch starts at once on the masked
        # stream, a helper launch on an ordinary stream waits for the previous step's gather and then pulls tiles from the same
        # queue on the units the gather has left.  No second pass over the frames, no guess at how long the gather takes.
        self.helper_cus = int(helper_cus)
        self._helper_stream = None
        self.helper_workgroups = None      # of the last split step (0: that launch was not split)
        # the gather runs whenever a process group exists -- with one rank too (bench.py --force-distributed puts the RCCL
        # path on a one-GPU box that way); without torch.distributed there is nothing to gather
        self.collective = dist.is_initialized()
        if self.collective and self.rank == dst:
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
            if self.collective:
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
        if self.helper_cus > 0:
            import torch
            cur = torch.cuda.current_stream()
            if self._helper_stream is None:
                self._helper_stream = torch.cuda.Stream()
            helper = self._helper_stream
            helper.wait_stream(cur)             # the frames, the fill, the previous step's launches
            prev = self._inflight.get((slot + 1) % len(labels_ring))       # the gather of step s-1: it holds the reserved CUs
            if prev is not None:
                with torch.cuda.stream(helper):
                    prev.wait()                 # (RCCL: the helper STREAM waits, the host goes on)
            self.helper_workgroups = self.ev.get_labels_forest_split(
                self.forest, depth, labels, helper.cuda_stream, self.helper_cus, labels_reduce=self.r, scale_factor=self.s,
                queue_tag=self._step_no & 1)
            cur.wait_stream(helper)             # the gather below (and the next step's launch) come after BOTH launches
        else:
            self.ev.get_labels_forest(self.forest, depth, labels, labels_reduce=self.r, scale_factor=self.s)
        if self.collective:
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
        """Rank `dst`: DeviceArray [world*frames, lh, lw] of every rank's labels (no process group: None)."""
        if self.gathered is None:
            return None
        return self.gathered.reshape(self.world * self.frames, self.lh, self.lw)


class PeerCopyGather:
    """Receive RING on rank `dst`, mapped into every process of the node.  Collective constructor.

    Layout of rank dst's allocation: n_slots x world x bytes_per_rank bytes of label maps -- step s of rank g lands in
    slot s % n_slots at [slot][g] -- followed by two arrays of uint64 counters, 64 bytes apart:
      ready[slot][g]    = s + 1 once rank g's copy of step s has landed (written by rank g with a second copy on the same
                          stream, i.e. after the data);
      consumed[slot]    = s + 1 once the consumer on rank dst has released step s (`release`).
    A consumer on rank dst calls wait_ready(s) (all world counters of the slot show s + 1), reads slot_array(s), then
    release(s); a producer that is about to reuse a slot waits for consumed[slot] >= s - n_slots + 1 first when flow control
    is on (PeerCopyForestEvaluator(flow_control=True)).  With two slots a consumer has one whole step to read.

    `ok` is False on every rank if any rank could not map the buffer (no IPC between the processes, e.g. devices hidden
    from each other) or if anything else went wrong on any rank -- `errors` then holds every rank's reason -- and callers
    fall back to the RCCL gather.  No rank raises out of the constructor before every collective has been entered by
    every rank."""

    FLAG_STRIDE = 64

    def __init__(self, world, rank, bytes_per_rank, dst=0, group=None, n_slots=2, _fail_open_on_rank=None):
        import torch
        import torch.distributed as dist
        self._lib = get_runtime().lib
        self.torch = torch
        self.world, self.rank, self.dst, self.bytes_per_rank = int(world), int(rank), int(dst), int(bytes_per_rank)
        self.n_slots = max(1, int(n_slots))
        self.data_bytes = self.n_slots * self.world * self.bytes_per_rank
        self.data_bytes_aligned = (self.data_bytes + 255) & ~255
        self.flag_bytes = self.n_slots * (self.world + 1) * self.FLAG_STRIDE
        self.base = None          # device pointer of the whole receive buffer as seen from THIS process
        self._owned, self._mapped = None, None
        self.errors = None
        # counters are read and written with small copies on a stream of their own (torch streams do not synchronise with the
        # default stream), so a poll never queues behind a forest launch; values come from a device table (index = value - 1)
        handle = [None]
        mine_ok, why = 1, None
        import threading
        self._host_lock = threading.Lock()     # the pinned landing buffer, the poll stream and the value table: one user at a time
        try:
            self._poll_stream = torch.cuda.Stream()
            self._host8 = torch.zeros(max(self.world + 1, 8) * self.n_slots * (self.FLAG_STRIDE // 8), dtype=torch.int64).pin_memory()
            self._vals_base, self._vals = 0, torch.arange(1, 65537, dtype=torch.int64, device="cuda")
            self._old_vals = []
            if self.rank == self.dst:
                p = ctypes.c_void_p()
                rc = self._lib.rdf_device_malloc(ctypes.byref(p), self.data_bytes_aligned + self.flag_bytes)
                buf = ctypes.create_string_buffer(64)
                if rc == 0:
                    self._owned = p.value
                    rc = self._lib.rdf_ipc_export(p, buf)
                if rc == 0:
                    handle[0] = bytes(buf.raw)
                    self.base = self._owned
                else:
                    mine_ok, why = 0, f"hipMalloc / hipIpcGetMemHandle: {self._err(rc)}"
        except Exception as e:      # noqa: BLE001 -- a rank that fails here must still reach the collectives below
            mine_ok, why = 0, repr(e)
        dist.broadcast_object_list(handle, src=self.dst, group=group)
        try:
            if self.rank != self.dst:
                if handle[0] is None:
                    mine_ok, why = 0, "rank dst exported no handle"
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
                        mine_ok, why = 0, f"hipIpcOpenMemHandle: {'refused by the test hook' if rc == -1 else self._err(rc)}"
        except Exception as e:      # noqa: BLE001
            mine_ok, why = 0, repr(e)

        def all_ok(v):
            flag = torch.tensor([v], dtype=torch.int32)
            if dist.get_backend(group) == "nccl":
                flag = flag.cuda()
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
            return int(flag.item()) == 1

        self.ok = all_ok(mine_ok)
        if self.ok:
            # end-to-end check before anything relies on it: every rank copies a pattern into its slot and raises its ready
            # counter, rank dst reads all of them back
            rc, n = -1, min(256, self.bytes_per_rank)
            try:
                if self.rank == self.dst:
                    self._flags_tensor().zero_()
                    torch.cuda.synchronize()
            except Exception as e:      # noqa: BLE001
                why = repr(e)
            zeroed = all_ok(1 if why is None else 0)      # (also a barrier: the counters are zero before anyone writes)
            try:
                if zeroed:
                    pattern = torch.full((n,), 17 + self.rank, dtype=torch.uint8, device="cuda")
                    st = torch.cuda.current_stream().cuda_stream
                    rc = self._lib.rdf_memcpy_device_async(ctypes.c_void_p(self.slot_ptr(self.rank)), ctypes.c_void_p(pattern.data_ptr()),
                                                           n, ctypes.c_void_p(st))
                    if rc == 0:
                        rc = self._signal(0, st, value=0xA5)
                    torch.cuda.synchronize()
            except Exception as e:      # noqa: BLE001
                rc, why = -1, repr(e)
            if rc != 0 and why is None:
                why = f"peer copy: {self._err(rc)}"
            copied = zeroed and all_ok(1 if rc == 0 else 0)      # (also a barrier: every pattern has landed)
            seen = 1
            try:
                if copied and self.rank == self.dst:
                    got = self.result_array().view(self.n_slots, self.world, self.bytes_per_rank)[0, :, :n].cpu()
                    want = (17 + torch.arange(self.world, dtype=torch.uint8)).view(-1, 1).expand(self.world, n)
                    flags = self.ready_counters()[0]
                    seen = 1 if bool((got == want).all()) and bool((flags == 0xA5).all()) else 0
                    if not seen:
                        why = "pattern or ready counters did not arrive"
                    self._flags_tensor().zero_()
                    torch.cuda.synchronize()
            except Exception as e:      # noqa: BLE001
                seen, why = 0, repr(e)
            self.ok = copied and all_ok(seen)
        reasons = [None] * self.world
        dist.all_gather_object(reasons, why, group=group)
        if not self.ok:
            self.errors = {g: r for g, r in enumerate(reasons) if r} or {"?": "a rank reported failure without a reason"}
            self.close()

    def _err(self, rc):
        msg = self._lib.rdf_error_string(int(rc))
        return (msg.decode() if isinstance(msg, bytes) else str(msg)) + f" (code {rc})"

    # -- layout ------------------------------------------------------------------------------------
    def slot_ptr(self, rank, byte_offset=0, step=0):
        return self.base + ((int(step) % self.n_slots) * self.world + rank) * self.bytes_per_rank + byte_offset

    def _ready_ptr(self, step, rank):
        return self.base + self.data_bytes_aligned + ((int(step) % self.n_slots) * (self.world + 1) + rank) * self.FLAG_STRIDE

    def _consumed_ptr(self, step):
        return self._ready_ptr(step, self.world)

    def _flags_tensor(self):
        """Rank dst: the counters as a torch int64 tensor [n_slots, world + 1, FLAG_STRIDE / 8] (zero copy)."""
        return self._view(self._owned + self.data_bytes_aligned, self.flag_bytes).view(self.torch.int64).view(
            self.n_slots, self.world + 1, self.FLAG_STRIDE // 8)

    def _view(self, ptr, nbytes):
        class _View:
            pass
        v = _View()
        v.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}
        return self.torch.as_tensor(v, device="cuda")

    def _value_ptr(self, value):
        """Device address of an int64 that holds `value` (>= 1): the source of a counter update by device-to-device copy."""
        value = int(value)
        with self._host_lock:
            return self._value_ptr_locked(value)

    def _value_ptr_locked(self, value):
        if not (self._vals_base < value <= self._vals_base + self._vals.numel()):
            self._old_vals = (self._old_vals + [self._vals])[-2:]      # (copies in flight may still read the old table)
            self._vals_base = ((value - 1) // 65536) * 65536
            self._vals = self.torch.arange(self._vals_base + 1, self._vals_base + 65537, dtype=self.torch.int64, device="cuda")
            self.torch.cuda.synchronize()
        return self._vals.data_ptr() + 8 * (value - 1 - self._vals_base)

    def _signal(self, step, stream, value=None):
        """Stream-ordered: this rank's ready counter of `step`'s slot := step + 1 (a second small copy behind the data)."""
        return self._lib.rdf_memcpy_device_async(ctypes.c_void_p(self._ready_ptr(step, self.rank)),
                                                 ctypes.c_void_p(self._value_ptr(int(step) + 1 if value is None else value)),
                                                 8, ctypes.c_void_p(stream))

    def _peek(self, ptr, n_words):
        """n_words int64 at device address `ptr`, read now (own stream), as a list."""
        # (the producer's wait_free and a consumer thread's wait_ready / ready_counters share the pinned landing buffer and
        # the poll stream: one reader at a time, the list is made before the next one may overwrite the buffer)
        with self._host_lock:
            st = ctypes.c_void_p(self._poll_stream.cuda_stream)
            rc = self._lib.rdf_memcpy_device_async(ctypes.c_void_p(self._host8.data_ptr()), ctypes.c_void_p(ptr), 8 * n_words, st)
            _lib.check(self._lib, rc, "rdf_memcpy_device_async (counters)")
            _lib.check(self._lib, self._lib.rdf_stream_synchronize(st), "rdf_stream_synchronize")
            return self._host8[:n_words].tolist()

    # -- producer side -----------------------------------------------------------------------------
    def push(self, step, src_ptr, nbytes, stream):
        """Copy this rank's label maps of `step` into its place in the ring and raise the ready counter behind them; both
        on `stream` (copy engines; nothing runs on a CU)."""
        rc = self._lib.rdf_memcpy_device_async(ctypes.c_void_p(self.slot_ptr(self.rank, 0, step)), ctypes.c_void_p(src_ptr), int(nbytes),
                                               ctypes.c_void_p(stream))
        _lib.check(self._lib, rc, "rdf_memcpy_device_async (label maps)")
        _lib.check(self._lib, self._signal(step, stream), "rdf_memcpy_device_async (ready counter)")

    def wait_free(self, step, timeout_s=30.0, poll_s=50e-6):
        """Producer: block (host) until the consumer has released the step that last used `step`'s slot."""
        import time
        need = int(step) - self.n_slots + 1
        if need <= 0:
            return True
        t0 = time.perf_counter()
        while True:
            if self._peek(self._consumed_ptr(step), 1)[0] >= need:
                return True
            if time.perf_counter() - t0 > timeout_s:
                return False
            time.sleep(poll_s)

    # -- consumer side (rank dst) ------------------------------------------------------------------
    def ready_counters(self):
        """Rank dst: int64 [n_slots, world] on the host."""
        words = self.FLAG_STRIDE // 8
        flat = self._peek(self._owned + self.data_bytes_aligned, self.n_slots * (self.world + 1) * words)
        t = self.torch.tensor(flat, dtype=self.torch.int64).view(self.n_slots, self.world + 1, words)
        return t[:, :self.world, 0]

    def is_ready(self, step):
        return bool((self.ready_counters()[int(step) % self.n_slots] >= int(step) + 1).all())

    def wait_ready(self, step, timeout_s=30.0, poll_s=50e-6):
        """Rank dst: block (host) until every rank's label maps of `step` have landed in the ring."""
        import time
        t0 = time.perf_counter()
        while not self.is_ready(step):
            if time.perf_counter() - t0 > timeout_s:
                return False
            time.sleep(poll_s)
        return True

    def release(self, step):
        """Rank dst: the consumer is done with `step`; its slot may be overwritten (by step + n_slots)."""
        st = ctypes.c_void_p(self._poll_stream.cuda_stream)
        rc = self._lib.rdf_memcpy_device_async(ctypes.c_void_p(self._consumed_ptr(step)), ctypes.c_void_p(self._value_ptr(int(step) + 1)), 8, st)
        _lib.check(self._lib, rc, "rdf_memcpy_device_async (consumed counter)")
        _lib.check(self._lib, self._lib.rdf_stream_synchronize(st), "rdf_stream_synchronize")

    def own_slot(self, step, shape, dtype=np.uint16):
        """This rank's place in the ring for `step` as a DeviceArray.  Rank dst evaluates straight into it instead of copying
        100 MB inside one device -- such a copy is a shader blit, not a copy-engine transfer, and takes CUs from the forest
        kernel (104 MB during a launch: 3.2 ms, the launch 4.40 instead of 4.04 ms on the test box).  On any other rank the
        array is rank dst's memory seen through the IPC mapping: a kernel that writes it stores across xGMI
        (PeerCopyForestEvaluator(direct_stores=True))."""
        assert self.base is not None and (self.rank == self.dst) == (self._owned is not None)
        nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        assert nbytes == self.bytes_per_rank
        st = _Storage.__new__(_Storage)
        st.rt, st.nbytes, st.version = get_runtime(), nbytes, 0
        st.handle = self._view(self.slot_ptr(self.rank, 0, step), nbytes)
        return DeviceArray(shape, dtype, st, 0)

    def slot_array(self, step):
        """Rank dst: step `step`'s label maps of all ranks, torch uint8 [world, bytes_per_rank] (zero copy)."""
        return self.result_array().view(self.n_slots, self.world, self.bytes_per_rank)[int(step) % self.n_slots]

    def result_array(self):
        """Rank `dst`: the whole ring as a torch uint8 tensor [n_slots * world * bytes_per_rank] (zero copy)."""
        return self._view(self._owned, self.data_bytes)

    def close(self):
        if self._mapped is not None:
            self._lib.rdf_ipc_close(ctypes.c_void_p(self._mapped))
            self._mapped = None
        if self._owned is not None:
            self._lib.rdf_device_free(ctypes.c_void_p(self._owned))
            self._owned = None
        self.base = None


class PeerCopyForestEvaluator:
    """One launch per step; each rank copies its label maps into rank `dst`'s ring on a side stream, so the copy
    of step s overlaps the launch of step s+1 (two label buffers used alternately, as in step_overlapped) -- or, with
    direct_stores, writes them there from the kernel itself.  With
    flow_control the producer waits (host side, before it reuses a ring slot) until the consumer on rank dst has released
    the step that used the slot before: PeerCopyGather.wait_ready / slot_array / release are the consumer's side."""

    def __init__(self, evaluator, forest, frames_per_rank, depth_dims, gather, labels_reduce=1, scale_factor=1., flow_control=False,
                 direct_stores=False):
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
        self.flow_control = bool(flow_control)
        # direct_stores: EVERY rank evaluates straight into its place in rank dst's ring -- the kernel's label stores cross
        # xGMI (a wave's 64 labels are one 128-byte store; 104 MB per 3.9-ms step = 27 GB/s per rank, one link each into
        # rank dst), there is no copy and no second label buffer.  The same thing with pinned host memory as the target
        # costs the kernel nothing (HostFramesEvaluator, profiles/r03_pcie_overlap.txt); across GPUs it has never run.
        self.direct_stores = bool(direct_stores)
        self._copied = {}
        self._own = {}            # (rank dst) DeviceArrays over its own places in the ring
        self.last_labels = None
        self._step_no = 0

    @property
    def steps_done(self):
        return self._step_no

    def step(self, depth, labels_ring, prefill=None):
        torch = self.torch
        g = self.gather
        cur = torch.cuda.current_stream()
        if g.rank == g.dst or self.direct_stores:
            # rank dst (with direct_stores: every rank) evaluates straight into its place in the ring (no copy); with flow
            # control it waits for the slot BEFORE the launch, not before a copy
            if self.flow_control and not g.wait_free(self._step_no):
                raise _lib.RdfError(f"rank {g.rank}: the consumer never released the ring slot of step {self._step_no}")
            key = self._step_no % g.n_slots
            if key not in self._own:
                # (pixels the kernel leaves untouched keep the caller's pre-fill: the ring's own places get the 65535 the
                # callers' buffers are created with, once)
                self._own[key] = g.own_slot(self._step_no, (self.frames, self.lh, self.lw)).fill(65535)
            labels = self._own[key]
            if prefill is not None:
                labels.fill(prefill)
            self.ev.get_labels_forest(self.forest, depth, labels, labels_reduce=self.r, scale_factor=self.s)
            done = torch.cuda.Event()
            done.record(cur)
            self.copy_stream.wait_event(done)
            _lib.check(self._lib, g._signal(self._step_no, self.copy_stream.cuda_stream), "rdf_memcpy_device_async (ready counter)")
            self.last_labels = labels
            self._step_no += 1
            return labels
        slot = self._step_no % len(labels_ring)
        labels = labels_ring[slot]
        assert tuple(labels.shape) == (self.frames, self.lh, self.lw)
        if self._copied.get(slot) is not None:
            cur.wait_event(self._copied[slot])      # this buffer's previous contents have left
        if prefill is not None:
            labels.fill(prefill)
        self.ev.get_labels_forest(self.forest, depth, labels, labels_reduce=self.r, scale_factor=self.s)
        done = torch.cuda.Event()
        done.record(cur)
        if self.flow_control and not self.gather.wait_free(self._step_no):
            raise _lib.RdfError(f"rank {self.gather.rank}: the consumer never released the ring slot of step {self._step_no}")
        self.copy_stream.wait_event(done)
        self.last_labels = labels
        self.gather.push(self._step_no, labels.ptr, self.nbytes, self.copy_stream.cuda_stream)
        ev = torch.cuda.Event()
        ev.record(self.copy_stream)
        self._copied[slot] = ev
        self._step_no += 1
        return labels

    def drain(self):
        """This rank's copies have landed.  (Rank dst may read after a barrier that follows every rank's drain.)"""
        self.copy_stream.synchronize()

    def result(self, step=None):
        """Rank dst: torch view [world*frames, lh, lw] (uint16 bits as int16) of one step's label maps -- the last step by
        default; else None."""
        if self.gather.rank != self.gather.dst:
            return None
        step = self._step_no - 1 if step is None else int(step)
        return self.gather.slot_array(max(step, 0)).view(self.torch.int16).view(self.gather.world * self.frames, self.lh, self.lw)
