"""The per-hand, per-frame chain of the reference's app, on one stream with no host round trip until the result.

Mirror of `App_3d_bz.run_per_hand_pipeline` (/root/reference/src/3d_bz.py:388-522):
    stencil the depth frame by hand group -> (flip x for the left hand) -> 0 -> 65535 ->
    LayeredDecisionForest.run -> (flip the labels back) -> RGBA debug image -> mean-shift modes ->
    fingertip heights above the calibrated plane.
Every step is a row of SURVEY 8 that this package already has (8a K, 8f-1, 8f-2); this class only owns the
intermediate buffers and the order.  The reference synchronises the context and reads the label means and the
depth frame back to the host for the last step (3d_bz.py:461-465, 503-522); here the means and the heights stay
on the device and ONE small copy (means + heights) ends the frame.
"""
import ctypes
import weakref

import numpy as np

from . import _lib
from .cuda.mean_shift import MeanShift
from .cuda.points_ops import PointsOps
from .device import DeviceArray, device_ptr, get_runtime
from .engine.buffer import GpuBuffer


# [graph, lib, capture id, device, last replay's event, capture stream, attempts] of replay objects that were dropped
_pending = []
_MAX_RELEASE_ATTEMPTS = 3


def _release_deferred():
    """Gives the tile-queue slots of dropped replay objects back (rdf_graph_slots_release) once their last replay has finished;
    does nothing while the current stream is being captured (the entries wait for the next call).  An entry leaves the list
    only when its slots HAVE been given back: one whose wait or reset raises is tried again at the next call (a few times --
    then it is dropped with its slots, which is what losing the entry at the first failure used to do silently)."""
    import torch
    try:
        if torch.cuda.is_current_stream_capturing():
            return
    except Exception:       # noqa: BLE001 -- (interpreter shutdown)
        return
    again = []
    while _pending:
        entry = _pending.pop()
        g, lib, cid, dev, event, stream, attempts = entry
        try:
            with torch.cuda.device(dev):         # (the slots are keyed by device: the device of the capture is made current)
                if event is not None:
                    event.synchronize()
                elif stream is not None:
                    # never replayed through replay() -- the raw graph may have been (replay.graph): the capture stream is
                    # where the warm-up ran and the best guess at where such a replay ran; the device, if there is none
                    stream.synchronize()
                else:
                    torch.cuda.synchronize()
                g.reset()
                lib.rdf_graph_slots_release(cid)
        except Exception:       # noqa: BLE001 -- a finalizer must not raise
            entry[6] = attempts + 1
            if entry[6] < _MAX_RELEASE_ATTEMPTS:
                again.append(entry)
    _pending.extend(again)


class HandPipeline:
    def __init__(self, layered_rdf, depth_dims, labels_reduce, eval_to_train_dim_ratio, mean_shift_rounds,
                 mean_shift_variances, fingertip_idxes, intrinsics, plane, depth_mm_level=0, fused_io=True):
        """depth_dims = (DIM_Y, DIM_X); intrinsics = (fx, fy, ppx, ppy) of the depth stream; plane = the calibrated
        plane's 4x4 matrix (calibrated_plane.plane); fingertip_idxes = 1-based composite label ids (3d_bz.py:112)."""
        self._rt = get_runtime()
        self._lib = self._rt.lib
        # run() writes the stack's per-layer label buffers: a stack that already serves another pipeline is replaced by a
        # sibling (same forests and tables, own label buffers), so that two pipelines -- the two hands of a frame -- can be
        # in flight together on two streams without overwriting each other's layer-0 labels.  `self.layered_rdf` is the
        # stack this pipeline really runs.
        # The mark is a weak reference: a stack whose pipeline is gone serves the next one itself again, and the mark keeps
        # nothing alive.
        owner = getattr(layered_rdf, "_pipeline_owner", None)
        if owner is not None and owner() is not None:
            layered_rdf = layered_rdf.sibling()
        layered_rdf._pipeline_owner = weakref.ref(self)
        self.layered_rdf = layered_rdf
        self.DIM_Y, self.DIM_X = int(depth_dims[0]), int(depth_dims[1])
        self.LABELS_REDUCE = int(labels_reduce)
        self.LABELS_DIM_Y, self.LABELS_DIM_X = self.DIM_Y // self.LABELS_REDUCE, self.DIM_X // self.LABELS_REDUCE
        self.EVAL_TO_TRAIN_DIM_RATIO = float(eval_to_train_dim_ratio)
        self.depth_mm_level = int(depth_mm_level)
        self.fused_io = bool(fused_io)   # False: the reference's nine launches around the forest, one by one
        self.mean_shift_rounds = int(mean_shift_rounds)
        self.fingertip_idxes = [int(i) for i in fingertip_idxes]
        self.intrinsics = tuple(float(v) for v in intrinsics)
        self.points_ops = PointsOps()
        self.mean_shift = MeanShift()
        d, l = (self.DIM_Y, self.DIM_X), (self.LABELS_DIM_Y, self.LABELS_DIM_X)
        self.depth_image_group = GpuBuffer(d, np.uint16)
        self.depth_image_2 = GpuBuffer(d, np.uint16)
        self.labels_image = GpuBuffer(l, np.uint16)
        self.labels_image_2 = GpuBuffer(l, np.uint16)
        self.labels_image_rgba = GpuBuffer(l + (4,), np.uint8)
        self.mean_shift_variances = DeviceArray((len(mean_shift_variances),), np.float32).set(
            np.asarray(mean_shift_variances, dtype=np.float32))
        self._ids = DeviceArray((len(self.fingertip_idxes),), np.int32).set(np.asarray(self.fingertip_idxes, np.int32))
        self._plane = DeviceArray((4, 4), np.float32).set(np.ascontiguousarray(plane, dtype=np.float32))
        # means [L,2] followed by heights [n_fingertips]: one device->host copy per frame
        self._L = int(layered_rdf.num_layered_classes)
        self._result = DeviceArray((self._L * 2 + len(self.fingertip_idxes),), np.float64)
        # fused_io: the modes and the heights are ONE launch (rdf_mean_shift_heights) that writes straight into pinned host
        # memory -- the frame ends with a stream synchronisation instead of a device-to-host copy
        self._host = None
        alloc = getattr(self._rt, "alloc_host_mapped", None)
        if self.fused_io and alloc is not None and len(self.fingertip_idxes) <= 1024:
            self._host = alloc(self._result.nbytes)

    def run(self, depth_image, depth_image_mm_groups, g_id, flip_x):
        """depth_image: GpuBuffer uint16 [DIM_Y, DIM_X] (the frame, 0 = no reading);
        depth_image_mm_groups: GpuBuffer of the hand-group image at mip level depth_mm_level;
        returns (label_means float64 [L, 2], fingertip heights float64 [n], NaN = "reset" in the reference)."""
        self._enqueue(depth_image, depth_image_mm_groups, g_id, flip_x)
        return self._read()

    def capture(self, depth_image, depth_image_mm_groups, g_id, flip_x):
        """Records the chain for these buffers and arguments into a hipGraph (through torch) and returns a
        function that replays it on the buffers' current contents and returns what run() returns: one graph
        launch per hand per frame instead of ~16 kernel launches.  replay(read=False) only enqueues (replay.read()
        fetches the result later), so the two hands of a frame -- two HandPipeline objects, each captured under its
        own torch stream -- can be in flight together.  Each pipeline owns every buffer its graph writes, the layered
        forest's per-layer label images included (__init__ takes a sibling of a stack that another pipeline already
        uses); the recorded forest launches get tile-queue slots of their own (rdf_hip.hip, g_sched), so a replay may
        run on any stream."""
        import torch
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):      # warm-up on the capture stream: workspaces, occupancy queries, queue slot
            self._enqueue(depth_image, depth_image_mm_groups, g_id, flip_x)
        side.synchronize()
        graph = torch.cuda.CUDAGraph()
        cap_id = ctypes.c_uint64(0)
        with torch.cuda.graph(graph, stream=side):
            named = self._lib.rdf_stream_capture_id(self._rt.stream(), ctypes.byref(cap_id)) == 0
            self._enqueue(depth_image, depth_image_mm_groups, g_id, flip_x)

        read_fn = self._read
        done = torch.cuda.Event()       # re-recorded behind every replay, on the stream the replay was launched on
        last = [None]

        def replay(read=True):
            graph.replay()
            done.record()
            last[0] = done
            return read_fn() if read else None
        replay.read = read_fn
        replay.graph = graph        # (replay() records the event the release waits for; a bare graph.replay() is not covered by it)
        # the recorded forest launches hold tile-queue slots of their own: given back when the replay object is dropped
        # (the graph goes with it), so that captures over a process's lifetime never run out of them
        if named:
            lib, cid, dev = self._lib, int(cap_id.value), torch.cuda.current_device()
            _release_deferred()      # (whatever an earlier finalizer had to put off)

            def _release(g=graph):
                # the last replay may still be running, on whatever stream it was launched on: its tile-queue slots must
                # not reach the next capture while its workgroups pull from them (two launches sharing a queue skip tiles).
                # Wait for THAT replay (its event), not for the device: the garbage collector may run this while the other
                # hand's pipeline is being captured, and a device-wide synchronisation is illegal during a capture (it
                # would invalidate that capture, and the error would be swallowed here).  While this thread's current
                # stream is capturing, the release is put off until the next capture() or replay release.
                _pending.append([g, lib, cid, dev, last[0], side, 0])
                _release_deferred()
            replay.release = weakref.finalize(replay, _release)
        return replay

    def _read(self):
        if self._host is not None:
            self._rt.synchronize()         # the current stream, as the copy below would be
            out = self._host[2].view(np.float64).copy()
        else:
            out = self._result.get()
        return out[:self._L * 2].reshape(self._L, 2), out[self._L * 2:]

    def _enqueue(self, depth_image, depth_image_mm_groups, g_id, flip_x):
        dims = np.array([self.DIM_X, self.DIM_Y], dtype=np.int32)
        ldims = np.array([self.LABELS_DIM_X, self.LABELS_DIM_Y], dtype=np.int32)
        po = self.points_ops
        if self.fused_io:
            # one pass in, none out: the stencil, the flip and 0 -> 65535 are one kernel, the flip back and the colouring
            # ride on the composite kernel's store (rdf_prepare_hand_depth, rdf_layered_run_hand)
            po.prepare_hand_depth(dims, np.int32(self.depth_mm_level), np.int32(g_id), depth_image_mm_groups.cu(),
                                  depth_image.cu(), self.depth_image_2.cu(), flip_x)
            self.layered_rdf.run_hand(self.depth_image_2, self.labels_image, self.EVAL_TO_TRAIN_DIM_RATIO, flip_x,
                                      self.labels_image_rgba)
        else:
            # the reference's sequence, kernel for kernel (3d_bz.py:396-456)
            self.depth_image_group.cu().fill(0)
            po.stencil_depth_image_by_group(dims, np.int32(self.depth_mm_level), np.int32(g_id), depth_image_mm_groups.cu(),
                                            depth_image.cu(), self.depth_image_group.cu())
            if flip_x:
                po.flip_x(dims, self.depth_image_group.cu(), self.depth_image_2.cu())
            else:
                self.depth_image_2.cu().copy_from(self.depth_image_group.cu())
            po.convert_0s_to_maxuint(np.int32(self.DIM_X * self.DIM_Y), self.depth_image_2.cu())

            self.layered_rdf.run(self.depth_image_2, self.labels_image, self.EVAL_TO_TRAIN_DIM_RATIO)

            if flip_x:
                self.labels_image_2.cu().copy_from(self.labels_image.cu())
                po.flip_x(ldims, self.labels_image_2.cu(), self.labels_image.cu())
            po.make_rgba_from_labels(np.uint32(self.LABELS_DIM_X), np.uint32(self.LABELS_DIM_Y), np.uint32(self._L),
                                     self.labels_image.cu(), self.layered_rdf.label_colors.cu(), self.labels_image_rgba.cu())

        if self._host is not None:
            base = self._host[1]
            self.mean_shift.run_device_with_heights(
                self.mean_shift_rounds, self.labels_image.cu().reshape((1, self.LABELS_DIM_Y, self.LABELS_DIM_X)), self._L,
                self.mean_shift_variances, self._ids, len(self.fingertip_idxes), depth_image.cu(), self.LABELS_REDUCE,
                self.intrinsics, self._plane, base, base + self._L * 2 * 8)
            return
        means = self.mean_shift.run_device(self.mean_shift_rounds,
                                           self.labels_image.cu().reshape((1, self.LABELS_DIM_Y, self.LABELS_DIM_X)),
                                           self._L, self.mean_shift_variances, out=self._result[:self._L * 2])
        fx, fy, ppx, ppy = self.intrinsics
        heights = self._result[self._L * 2:]
        # z is looked up in the ORIGINAL depth frame (3d_bz.py:515), not the stencilled / flipped one
        rc = self._lib.rdf_fingertip_heights(means.ptr, self._L, self._ids.ptr, len(self.fingertip_idxes),
                                             device_ptr(depth_image.cu()), self.DIM_X, self.DIM_Y, self.LABELS_REDUCE,
                                             fx, fy, ppx, ppy, self._plane.ptr, heights.ptr, self._rt.stream())
        _lib.check(self._lib, rc, "rdf_fingertip_heights")
        self._result.mark_dirty()
