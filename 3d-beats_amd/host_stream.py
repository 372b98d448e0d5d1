"""Depth frames that live on the HOST: labels at the kernel's pace, transfers included.

The reference uploads every frame and reads every result back on one stream (/root/reference/src/run_live_layered.py:66-81,
`GpuBuffer.cu().set(...)` / `.get()`): transfer, kernel and transfer follow each other -- on MI355X 1.9 + 3.85 + 1.8 ms for a
batch of 128 frames of 848x480 (104 MB each way over PCIe gen 5), 48 % of the HBM-resident rate.

`HostFramesEvaluator` runs the two transfers beside the kernel instead:
  * frames go up on a copy engine (one stream of its own) into one of two device slots while the kernel evaluates the other;
  * labels are WRITTEN TO THE HOST BY THE KERNEL: the label slot is pinned host memory mapped into the device's address space,
    and a wave's 64 labels are one 128-byte store that crosses PCIe as it is.  The batch's kernel takes the same 3.85 ms
    with its labels in host memory as in HBM (the stores are posted and 27 GB/s is half of what the link takes), so
    there is no download stage at all -- and the link carries frames up and labels down at the same time, which two copy
    engine transfers did not (H2D next to D2H took the sum of both, 3.7 ms: tools/pcie_overlap_probe.py).
Measured (tools/pcie_overlap_probe.py --host-labels, profiles/r03_pcie_overlap.txt): 3.93-3.97 ms per step, every step
(13.2 Gpix/s, the HBM-resident rate); with the labels downloaded by a copy engine (`labels_by_copy_engine=True`, two more
device slots and a third stream) the best split (two pieces) has a median step of 4.1-4.2 ms but steps stall behind the other
direction's transfer (mean 4.8-5.3 ms, worst 9.6).

Everything is enqueued without blocking the host; `result()` waits for one step's labels.

    p = HostFramesEvaluator(forest, (128, 480, 848))
    p.next_frames()[:] = batch0; t0 = p.submit()
    p.next_frames()[:] = batch1; t1 = p.submit()      # batch0 is being evaluated
    labels0 = p.result(t0)                            # valid until the second submit() after this one
"""
import numpy as np

from .decision_tree import DecisionTreeEvaluator
from .device import DeviceArray, get_runtime, host_mapped_array

N_SLOTS = 2


class HostFramesEvaluator:
    def __init__(self, forest, frames_shape, labels_reduce=1, scale_factor=1., evaluator=None, pieces=1,
                 labels_by_copy_engine=False):
        """forest: DecisionForest; frames_shape = (frames per step, DIM_Y, DIM_X), the same for every step.  `pieces`: parts
        a step is cut into (upload of a part overlaps the evaluation of the part before it: a shorter first-result
        latency; one part is the fastest in steady state)."""
        self._rt = get_runtime()
        torch = self._torch = self._rt.torch
        self.forest = forest
        self.evaluator = evaluator if evaluator is not None else DecisionTreeEvaluator()
        self.labels_reduce, self.scale_factor = int(labels_reduce), float(scale_factor)
        f, h, w = (int(x) for x in frames_shape)
        self.frames_shape = (f, h, w)
        self.labels_shape = (f, h // self.labels_reduce, w // self.labels_reduce)
        self.pieces = max(1, min(int(pieces), f))
        self.labels_by_copy_engine = bool(labels_by_copy_engine)
        self._cuts = [(f * c) // self.pieces for c in range(self.pieces + 1)]
        n_in, n_out = f * h * w, int(np.prod(self.labels_shape))
        self._per_in, self._per_out = h * w, self.labels_shape[1] * self.labels_shape[2]
        # frames: pinned host slots (numpy views of pinned torch tensors) and device slots
        self._in_t = [torch.empty(n_in, dtype=torch.int16).pin_memory() for _ in range(N_SLOTS)]
        self.frames = [t.numpy().view(np.uint16).reshape(self.frames_shape) for t in self._in_t]
        self._depth = [DeviceArray(self.frames_shape, np.uint16) for _ in range(N_SLOTS)]
        self._depth_t = [d.torch_bytes().view(torch.int16) for d in self._depth]
        self._s_up = torch.cuda.Stream()
        # labels: host slots the kernel writes, or device slots + pinned host slots + a download stream
        if self.labels_by_copy_engine:
            self._out_t = [torch.empty(n_out, dtype=torch.int16).pin_memory() for _ in range(N_SLOTS)]
            self.labels = [t.numpy().view(np.uint16).reshape(self.labels_shape) for t in self._out_t]
            self._labels_dev = [DeviceArray(self.labels_shape, np.uint16) for _ in range(N_SLOTS)]
            self._labels_t = [l.torch_bytes().view(torch.int16) for l in self._labels_dev]
            self._s_dn = torch.cuda.Stream()
        else:
            pairs = [host_mapped_array(self.labels_shape, np.uint16) for _ in range(N_SLOTS)]
            self._labels_dev = [p[0] for p in pairs]
            self.labels = [p[1] for p in pairs]
        # per slot: events of its last use (None before the first)
        self._uploaded = [None] * N_SLOTS        # the slot's frames have left the pinned host array
        self._evaluated = [[None] * self.pieces for _ in range(N_SLOTS)]
        self._arrived = [[None] * self.pieces for _ in range(N_SLOTS)]     # the part's labels are in the host slot
        self._step = 0
        self.step_marks = []          # (mark_steps) one timing event per submit on the evaluate stream
        self.mark_steps = False

    # -- the caller's side -------------------------------------------------------------------------------------
    def next_frames(self):
        """The pinned host array the next submit() uploads: (frames, DIM_Y, DIM_X) uint16.  Blocks only if the upload of
        the step that used this slot two submits ago has not finished."""
        b = self._step % N_SLOTS
        if self._uploaded[b] is not None:
            self._uploaded[b].synchronize()
        return self.frames[b]

    def submit(self):
        """Enqueue upload -> evaluate (-> download) of the frames in next_frames(); returns the step's ticket.  The label
        slot of the step two submits ago is overwritten."""
        torch = self._torch
        s, b = self._step, self._step % N_SLOTS
        cur = torch.cuda.current_stream()
        if self.mark_steps:
            e = torch.cuda.Event(enable_timing=True)
            e.record(cur)
            self.step_marks.append(e)
        for c in range(self.pieces):
            a0, a1 = self._cuts[c], self._cuts[c + 1]
            with torch.cuda.stream(self._s_up):
                if self._evaluated[b][c] is not None:        # the kernel that read this part two steps ago
                    self._s_up.wait_event(self._evaluated[b][c])
                self._depth_t[b][a0 * self._per_in:a1 * self._per_in].copy_(
                    self._in_t[b][a0 * self._per_in:a1 * self._per_in], non_blocking=True)
                up = torch.cuda.Event()
                up.record(self._s_up)
            cur.wait_event(up)
            if self.labels_by_copy_engine and self._arrived[b][c] is not None:   # the device slot's previous labels are on the host
                cur.wait_event(self._arrived[b][c])
            # the evaluation leaves pixels without depth alone (tree_eval.cu:81-89): they become 65535 in the same pass, the
            # fill of LayeredDecisionForest.run (decision_tree.py:237-240) folded in
            self.evaluator.get_labels_forest_filled(self.forest, self._depth[b][a0:a1], self._labels_dev[b][a0:a1],
                                                    self.labels_reduce, None, None, self.scale_factor)
            ev = torch.cuda.Event()
            ev.record(cur)
            self._evaluated[b][c] = ev
            if self.labels_by_copy_engine:
                with torch.cuda.stream(self._s_dn):
                    self._s_dn.wait_event(ev)
                    self._out_t[b][a0 * self._per_out:a1 * self._per_out].copy_(
                        self._labels_t[b][a0 * self._per_out:a1 * self._per_out], non_blocking=True)
                    dn = torch.cuda.Event()
                    dn.record(self._s_dn)
                    self._arrived[b][c] = dn
            else:
                self._arrived[b][c] = ev      # the kernel's own stores: on the host when the kernel has finished
        self._uploaded[b] = up
        self._step += 1
        return s

    def result(self, ticket):
        """Labels of step `ticket` as a pinned host array (labels_shape, uint16); blocks until they have arrived.  The array
        is the slot's own: it is overwritten by the second submit() after `ticket`."""
        if not (self._step - N_SLOTS <= ticket < self._step):
            raise ValueError(f"step {ticket} is not in flight (next step {self._step}, {N_SLOTS} slots)")
        b = ticket % N_SLOTS
        for e in self._arrived[b]:
            e.synchronize()
        return self.labels[b]

    def drain(self):
        """Block until everything submitted so far has arrived on the host."""
        for b in range(N_SLOTS):
            for e in self._arrived[b]:
                if e is not None:
                    e.synchronize()
