"""HostFramesEvaluator: frames in pinned host memory in, labels in pinned host memory out, as a pipeline over two slots --
the labels written to the host by the kernel itself (the default) or downloaded by a copy engine.
The reference's counterpart is the per-frame upload / evaluate / read back of run_live_layered.py:66-81; every step's labels
must be the oracle's for THAT step's frames, whatever is in flight around it."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("copy_engine", [False, True])
@pytest.mark.parametrize("pieces,reduce", [(2, 1), (1, 1), (3, 2), (8, 1)])
def test_every_step_gets_its_own_frames_labels(pieces, reduce, copy_engine, rdf, gpu_runtime, oracle):
    synth = rdf.synth
    n, h, w = 6, 96, 168
    f_np = synth.forest(4, 9, 4, "trained", 80)
    forest = rdf.DecisionForest.from_numpy(f_np)
    p = rdf.HostFramesEvaluator(forest, (n, h, w), labels_reduce=reduce, pieces=pieces, labels_by_copy_engine=copy_engine)
    assert p.next_frames().shape == (n, h, w) and p.labels_shape == (n, h // reduce, w // reduce)
    steps = 7
    batches = [synth.frames(["dense", "live"] * (n // 2), 500 + 10 * s, h, w) for s in range(steps)]
    for b in batches:
        b[:, :3, :5] = 0                      # pixels without depth stay 65535 in every step (the slot is refilled)
    want = []
    for b in batches:
        x = np.full(p.labels_shape, 65535, np.uint16)
        oracle.eval_forest(b, f_np, x, reduce)
        want.append(x)
    tickets, got = [], {}
    for s in range(steps):
        p.next_frames()[:] = batches[s]
        tickets.append(p.submit())
        if s >= 1:                            # read one step behind: two steps are in flight
            got[s - 1] = p.result(tickets[s - 1]).copy()
    got[steps - 1] = p.result(tickets[-1]).copy()
    p.drain()
    for s in range(steps):
        assert np.array_equal(got[s], want[s]), (s, int((got[s] != want[s]).sum()))
    # a ticket whose slot has been reused is refused, not answered with another step's labels
    with pytest.raises(ValueError):
        p.result(tickets[0])
    with pytest.raises(ValueError):
        p.result(steps)


def test_next_frames_waits_for_the_slots_upload(rdf, gpu_runtime, oracle):
    """Overwriting a host slot right after its step was submitted must not change that step's labels: next_frames() hands
    the slot out again only when its upload has left."""
    synth = rdf.synth
    n, h, w = 4, 120, 200
    f_np = synth.forest(3, 8, 4, "full", 90)
    forest = rdf.DecisionForest.from_numpy(f_np)
    p = rdf.HostFramesEvaluator(forest, (n, h, w))
    a, b, c = (synth.frames(["dense"] * n, 900 + k, h, w) for k in range(3))
    p.next_frames()[:] = a
    t0 = p.submit()
    p.next_frames()[:] = b
    t1 = p.submit()
    buf = p.next_frames()                     # slot of step 0 again: blocks until step 0's frames are on the device
    buf[:] = c
    got0 = p.result(t0).copy()
    t2 = p.submit()
    want = [np.full((n, h, w), 65535, np.uint16) for _ in range(3)]
    for x, fr in zip(want, (a, b, c)):
        oracle.eval_forest(fr, f_np, x)
    assert np.array_equal(got0, want[0])
    assert np.array_equal(p.result(t1), want[1])
    assert np.array_equal(p.result(t2), want[2])


def test_layered_run_writes_its_composite_into_host_memory(rdf, gpu_runtime, oracle):
    """A result buffer in pinned, device-mapped host memory (GpuBuffer(..., host_mapped=True)): LayeredDecisionForest.run's
    composite lands in `.host` with no read back, the same image as in a device buffer (both routes of run())."""
    synth = rdf.synth
    h, w, r = 240, 424, 2
    f0, f1 = synth.forest(3, 9, 4, "trained", 60), synth.forest(3, 10, 5, "trained", 70)
    conditions = [[0, 1], [0, 2], [1, 3], [0, 3], [0, 4], [0, 5], [0, 6], [0, 7]]
    cfg = {"layers": [{"model": rdf.DecisionForest.from_numpy(f0)},
                      {"model": rdf.DecisionForest.from_numpy(f1), "filter_model": 0, "filter_model_class": 3}],
           "conditions": conditions, "label_colors": [[10 * i, 0, 0, 255] for i in range(1, 8)]}
    frame = synth.frames(["live"], 4100, h, w)[0]
    l0 = np.full((1, h // r, w // r), 65535, np.uint16)
    l1, comp = l0.copy(), l0.copy()
    oracle.eval_forest(frame[None], f0, l0, r)
    oracle.eval_forest(frame[None], f1, l1, r, l0, 3)
    oracle.composite([l0[0], l1[0]], np.array(conditions, np.int32), comp)
    dbuf = rdf.GpuBuffer((h, w), np.uint16)
    dbuf.cu().set(frame)
    for fused in (True, False):
        lf = rdf.LayeredDecisionForest(cfg, (h, w), r, fused=fused)
        out = rdf.GpuBuffer((h // r, w // r), np.uint16, host_mapped=True)
        assert out.host is not None and out.host.shape == (h // r, w // r)
        out.host[:] = 7
        lf.run(dbuf, out, 1.0)
        gpu_runtime.synchronize()
        assert np.array_equal(out.host, comp[0]), (fused, int((out.host != comp[0]).sum()))


def test_host_mapped_get_set_and_fill_wait_for_the_stream(rdf, gpu_runtime, oracle):
    """The reference idiom `labels.cu().get()` right after a launch (run_live.py:131) on a host-mapped buffer: get(), set()
    and byte fills are host-memory accesses, so they wait for the stream that may still be writing or reading the memory --
    no manual synchronize."""
    synth = rdf.synth
    h, w = 480, 848
    forest_np = synth.forest(4, 16, 4, "full", 11)
    forest = rdf.DecisionForest.from_numpy(forest_np)
    frames = synth.frames(["dense", "dense", "live", "dense"], 4300, h, w)
    want = np.full(frames.shape, 65535, np.uint16)
    oracle.eval_forest(frames, forest_np, want)
    ev = rdf.DecisionTreeEvaluator()
    depth = rdf.to_device(frames)
    labels, host = rdf.host_mapped_array(frames.shape, np.uint16)
    for rep in range(3):
        labels.fill(65535)                      # (a u16 fill is a kernel on the stream)
        ev.get_labels_forest(forest, depth, labels)
        got = labels.get()                      # no synchronize in between
        assert np.array_equal(got, want), (rep, int((got != want).sum()))
    # set() right behind a launch that READS the memory: a host-mapped DEPTH buffer
    dmap, _ = rdf.host_mapped_array(frames.shape, np.uint16)
    dmap.set(frames)
    out = rdf.DeviceArray(frames.shape, np.uint16).fill(65535)
    ev.get_labels_forest(forest, dmap, out)
    dmap.set(np.zeros_like(frames))             # must not reach the kernel that is still reading the frames
    assert np.array_equal(out.get(), want)
    # a byte fill (not a u16 pattern) behind a launch that writes
    lab8, host8 = rdf.host_mapped_array((16,), np.uint8)
    lab8.fill(np.uint8(3))
    assert (host8 == 3).all()
