"""Host logic of 3d-beats_amd/decision_tree.py with the host-memory test double (no GPU):
reference-compatible surface, shape asserts, layer wiring, pack caching, batch chunking."""
import json
import os

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "rdf_golden_v1.npz")


def test_get_config_matches_reference_formula(rdf):
    # decision_tree.py:135-144; values confirmed against the reference module in SURVEY 8c
    assert rdf.DecisionTree.get_config(20, 4) == (1048575, 1048576, 15)
    assert rdf.DecisionTree.get_config(10, 3) == (1023, 1024, 13)


def test_public_surface_names(rdf):
    dt = __import__("importlib").import_module("3d-beats_amd.decision_tree")
    for cls, members in {
        "DecisionTree": ["get_config"],
        "DecisionForest": ["load"],
        "LayeredDecisionForest": ["load", "run"],
        "DecisionTreeEvaluator": ["get_labels", "get_labels_forest", "make_composite_labels_image"],
    }.items():
        for m in members:
            assert hasattr(getattr(dt, cls), m), f"{cls}.{m}"
    import inspect
    sig = inspect.signature(dt.DecisionTreeEvaluator.get_labels_forest)
    assert list(sig.parameters) == ["self", "forest", "depth_images_in", "labels_out", "labels_reduce",
                                    "filter_images", "filter_images_class", "scale_factor"]
    assert sig.parameters["labels_reduce"].default == 1 and sig.parameters["scale_factor"].default == 1.
    sig = inspect.signature(dt.LayeredDecisionForest.run)
    assert list(sig.parameters) == ["self", "depth_image", "labels_image", "scale_factor"]
    sig = inspect.signature(dt.LayeredDecisionForest.load)
    assert list(sig.parameters) == ["config_filename", "depth_dims", "labels_reduce"]


def test_device_array_basics(rdf, host_runtime):
    a = rdf.DeviceArray((2, 3, 4), np.uint16)
    a.fill(65535)
    assert (a.get() == 65535).all()
    v0 = a.version
    a[1].set(np.arange(12, dtype=np.uint16).reshape(3, 4))
    assert a.version > v0
    assert np.array_equal(a.get()[1], np.arange(12).reshape(3, 4))
    assert (a.get()[0] == 65535).all()
    assert a.reshape(6, 4).shape == (6, 4) and a.reshape((24,)).ptr == a.ptr
    assert a[1].ptr == a.ptr + 24
    f = rdf.DeviceArray((5,), np.float32).fill(np.float32(1.5))
    assert (f.get() == 1.5).all()
    i = rdf.DeviceArray((3,), np.int64).fill(-2)
    assert (i.get() == -2).all()
    assert rdf.device_ptr(a) == a.ptr == a.__cuda_array_interface__["data"][0]
    with pytest.raises(AssertionError):
        a.set(np.zeros((2, 3, 4), np.int16))


def test_forest_load_and_flat_eval(rdf, host_runtime, tmp_path):
    g = np.load(GOLDEN)
    p = tmp_path / "m.npy"
    np.save(p, g["g1_forest"])
    forest = rdf.DecisionForest.load(str(p))
    assert (forest.num_trees, forest.max_depth, forest.num_classes) == (4, 8, 4)
    assert forest.forest_cu.shape == g["g1_forest"].shape
    ev = rdf.DecisionTreeEvaluator()
    depth = rdf.to_device(g["g1_depth"])
    labels = rdf.DeviceArray(g["g1_labels"].shape, np.uint16).fill(65535)
    ev.get_labels_forest(forest, depth, labels)
    assert np.array_equal(labels.get(), g["g1_labels"])
    # shape asserts of decision_tree.py:301-305
    with pytest.raises(AssertionError):
        ev.get_labels_forest(forest, depth, rdf.DeviceArray((2, 24, 32), np.uint16))
    with pytest.raises(AssertionError):
        ev.get_labels_forest(forest, depth, labels, filter_images=labels)  # class missing
    with pytest.raises(AssertionError):
        ev.get_labels_forest(forest, depth, labels, filter_images=rdf.DeviceArray((1, 2, 2), np.uint16),
                             filter_images_class=1)


def test_pack_cache_follows_forest_writes_and_scale(rdf, host_runtime):
    g = np.load(GOLDEN)
    forest = rdf.DecisionForest.from_numpy(g["g2_forest"])
    ev = rdf.DecisionTreeEvaluator()
    depth = rdf.to_device(g["g2_depth"])
    filt = rdf.to_device(g["g2_filter"])
    labels = rdf.DeviceArray(g["g2_labels"].shape, np.uint16).fill(0)
    calls = host_runtime.lib.calls

    def packs():
        return [c for c in calls if c[0] == "rdf_forest_pack"]

    ev.get_labels_forest(forest, depth, labels, 2, filt, 1, 0.5)
    assert np.array_equal(labels.get(), g["g2_labels"])
    assert len(packs()) == 1 and packs()[0][-1] == 0.5
    ev.get_labels_forest(forest, depth, labels, 2, filt, 1, 0.5)
    assert len(packs()) == 1                       # cached
    ev.get_labels_forest(forest, depth, labels, 2, filt, 1, 1.0)
    assert len(packs()) == 2                       # new scale
    forest.forest_cu[0].set(g["g2_forest"][0])     # write through a view => repack
    ev.get_labels_forest(forest, depth, labels, 2, filt, 1, 0.5)
    assert len(packs()) == 3
    # unpacked evaluator goes straight to rdf_eval_forest
    ev2 = rdf.DecisionTreeEvaluator(use_packed=False)
    labels.fill(0)
    ev2.get_labels_forest(forest, depth, labels, 2, filt, 1, 0.5)
    assert calls[-1][0] == "rdf_eval_forest" and np.array_equal(labels.get(), g["g2_labels"])


def test_single_tree_get_labels(rdf, host_runtime):
    g = np.load(GOLDEN)
    t = rdf.DecisionTree(7, 6)
    assert t.tree_out_cu.shape == (127, 19) and (t.tree_out_cu.get() == 0).all()
    t.tree_out_cu.set(g["g4_tree"])
    ev = rdf.DecisionTreeEvaluator()
    out = rdf.DeviceArray(g["g4_labels"].shape, np.uint16).fill(7)
    ev.get_labels(t, rdf.to_device(g["g4_depth"]), out)
    assert np.array_equal(out.get(), g["g4_labels"])


def _write_layered_cfg(tmp_path, g):
    np.save(tmp_path / "l0.npy", g["g3_forest0"])
    np.save(tmp_path / "l1.npy", g["g3_forest1"])
    cfg = {"layers": [{"model": "l0.npy"}, {"model": "l1.npy", "filter_model": 0, "filter_model_class": 3}],
           "conditions": g["g3_cond"].tolist(),
           "label_colors": [[255, 0, 0, 255], [0, 255, 0, 255], [0, 0, 255, 255], [255, 255, 0, 255]]}
    p = tmp_path / "model_cfg.json"
    p.write_text(json.dumps(cfg))
    return str(p)


def test_layered_forest_run_matches_golden(rdf, host_runtime, tmp_path):
    """The call convention of run_live_layered.py:54-58,126 / 3d_bz.py:389-437."""
    g = np.load(GOLDEN)
    cfg = _write_layered_cfg(tmp_path, g)
    lf = rdf.LayeredDecisionForest.load(cfg, (60, 84), labels_reduce=2)
    lf.fused = False   # the reference's step-by-step sequence; the one-call path is tested below
    assert lf.labels_dims == (30, 42) and lf.num_models == 2 and lf.num_layered_classes == 4
    assert lf.label_colors.shape == (4, 4) and len(lf.label_images) == 2
    depth = rdf.GpuBuffer((60, 84), np.uint16)
    depth.cu().set(g["g3_depth"][0])
    labels = rdf.GpuBuffer((30, 42), np.uint16)
    labels.cu().fill(123)                      # run() must pre-fill with 65535 itself
    lf.run(depth, labels, 1.0)
    assert np.array_equal(lf.label_images[0].cu().get(), g["g3_l0"][0])
    assert np.array_equal(lf.label_images[1].cu().get(), g["g3_l1"][0])
    assert np.array_equal(labels.cu().get(), g["g3_comp"][0])
    assert lf.eval.composite_bad_pixels() == 0
    # order of work: 3 fills, layer 0, layer 1 (filter class 3), composite (decision_tree.py:237-264)
    names = [c[0] for c in host_runtime.lib.calls if c[0] != "rdf_forest_pack"]
    assert names[-6:] == ["rdf_fill_u16"] * 3 + ["rdf_eval_forest_packed"] * 2 + ["rdf_composite"]
    evals = [c for c in host_runtime.lib.calls if c[0] == "rdf_eval_forest_packed"]
    assert evals[-2][7] == -1 and evals[-1][7] == 3 and evals[-1][8] == 2


def test_layered_forest_fused_run_equals_stepwise(rdf, host_runtime, tmp_path):
    g = np.load(GOLDEN)
    lf = rdf.LayeredDecisionForest.load(_write_layered_cfg(tmp_path, g), (60, 84), labels_reduce=2)
    assert lf.fused
    depth, labels = rdf.GpuBuffer((60, 84), np.uint16), rdf.GpuBuffer((30, 42), np.uint16)
    depth.cu().set(g["g3_depth"][0])
    labels.cu().fill(7)
    lf.run(depth, labels, 1.0)
    assert [c[0] for c in host_runtime.lib.calls if c[0].startswith("rdf_layered")] == ["rdf_layered_run"]
    assert np.array_equal(lf.label_images[0].cu().get(), g["g3_l0"][0])
    assert np.array_equal(lf.label_images[1].cu().get(), g["g3_l1"][0])
    assert np.array_equal(labels.cu().get(), g["g3_comp"][0])


def test_layer_filtering_on_a_later_layer_takes_the_reference_sequence(rdf, host_runtime, tmp_path):
    """decision_tree.py:237-257 fills every label buffer with 65535 BEFORE the first layer runs, so a layer whose
    filter_model is itself or a later layer sees only 65535 and classifies nothing -- frame after frame.  The fused call
    folds the fills into the kernels (a later layer's buffer would still hold the previous frame), so such stacks must take
    the step-by-step path."""
    g = np.load(GOLDEN)
    cfg = json.loads(open(_write_layered_cfg(tmp_path, g)).read())
    cfg["root"] = str(tmp_path)
    cfg["layers"][0]["filter_model"], cfg["layers"][0]["filter_model_class"] = 1, 2    # layer 0 filtered on layer 1
    lf = rdf.LayeredDecisionForest(cfg, (60, 84), 2)
    assert not lf.fused
    depth, labels = rdf.GpuBuffer((60, 84), np.uint16), rdf.GpuBuffer((30, 42), np.uint16)
    depth.cu().set(g["g3_depth"][0])
    for _ in range(2):          # the second frame must not see the first frame's layer-1 labels
        lf.run(depth, labels, 1.0)
        assert (lf.label_images[0].cu().get() == 65535).all()
        assert (lf.label_images[1].cu().get() == 65535).all()     # filtered on class 3 of an all-65535 layer 0
        assert (labels.cu().get() == 65535).all()
    assert not any(c[0] == "rdf_layered_run" for c in host_runtime.lib.calls)


def test_sibling_stack_shares_forests_but_not_label_buffers(rdf, host_runtime, tmp_path):
    g = np.load(GOLDEN)
    lf = rdf.LayeredDecisionForest.load(_write_layered_cfg(tmp_path, g), (60, 84), labels_reduce=2)
    sib = lf.sibling()
    assert [m for m, _, _ in sib.m] == [m for m, _, _ in lf.m] and [x[1:] for x in sib.m] == [x[1:] for x in lf.m]
    assert all(a is not b for a, b in zip(sib.label_images, lf.label_images)) and sib.eval is not lf.eval
    assert sib.num_layered_classes == lf.num_layered_classes and sib.labels_dims == lf.labels_dims
    assert np.array_equal(sib.label_colors.cu().get(), lf.label_colors.cu().get())
    depth, la, lb = rdf.GpuBuffer((60, 84), np.uint16), rdf.GpuBuffer((30, 42), np.uint16), rdf.GpuBuffer((30, 42), np.uint16)
    depth.cu().set(g["g3_depth"][0])
    lf.run(depth, la, 1.0)
    lf.label_images[0].cu().fill(9)        # scribbling over one stack's layer buffers does not reach the other's
    sib.run(depth, lb, 1.0)
    assert np.array_equal(lb.cu().get(), g["g3_comp"][0]) and np.array_equal(la.cu().get(), g["g3_comp"][0])
    assert np.array_equal(sib.label_images[0].cu().get(), g["g3_l0"][0]) and (lf.label_images[0].cu().get() == 9).all()


def test_layered_cfg_validation(rdf, host_runtime, tmp_path):
    g = np.load(GOLDEN)
    cfg = json.loads(open(_write_layered_cfg(tmp_path, g)).read())
    cfg["root"] = str(tmp_path)
    cfg["label_colors"] = cfg["label_colors"][:3]
    with pytest.raises(AssertionError):          # decision_tree.py:228
        rdf.LayeredDecisionForest(cfg, (60, 84), 2)
    cfg = json.loads(open(_write_layered_cfg(tmp_path, g)).read())
    cfg["root"] = str(tmp_path)
    del cfg["layers"][1]["filter_model_class"]
    with pytest.raises(KeyError):                # decision_tree.py:192-194
        rdf.LayeredDecisionForest(cfg, (60, 84), 2)


def test_composite_bad_pixels_are_counted_not_fatal(rdf, host_runtime):
    ev = rdf.DecisionTreeEvaluator()
    img = rdf.to_device(np.array([[1, 2, 3]], dtype=np.uint16))
    table = rdf.to_device(np.array([rdf.device_ptr(img)], dtype=np.int64))
    cond = rdf.to_device(np.array([(0, 4), (1, 0)], dtype=np.int32))
    out = rdf.DeviceArray((1, 1, 3), np.uint16).fill(65535)
    ev.make_composite_labels_image(table, 3, 1, cond, out)
    assert out.get().tolist() == [[[4, 65535, 65535]]]
    assert ev.composite_bad_pixels() == 2 and ev.composite_bad_pixels() == 0


def test_batches_beyond_2_31_pixels_are_chunked(rdf):
    dt = __import__("importlib").import_module("3d-beats_amd.decision_tree")
    chunks = dt._image_chunks(6000, 848 * 480)   # 2.44e9 pixels
    assert sum(n for _, n in chunks) == 6000 and len(chunks) == 2
    assert all(n * 848 * 480 < 2 ** 31 for _, n in chunks)
    assert chunks[0][0] == 0 and chunks[1][0] == chunks[0][1]
    assert dt._image_chunks(1, 848 * 480) == [(0, 1)]


def test_config1_single_frame_plumbing_on_cpu(rdf, host_runtime, oracle, oracle_np):
    """BASELINE config 1: one live-like 848x480 frame, 1 tree of depth 10, no GPU -- the reference-shaped API
    end to end on the host-memory test double (the product itself has no CPU backend), C and numpy
    restatements agreeing on the full-size frame."""
    synth = rdf.synth
    forest_np = synth.forest(1, 10, 4, "full")
    frame = synth.frames(["live"], 0)
    forest = rdf.DecisionForest.from_numpy(forest_np)
    ev = rdf.DecisionTreeEvaluator()
    labels = rdf.DeviceArray((1, 480, 848), np.uint16).fill(65535)
    ev.get_labels_forest(forest, rdf.to_device(frame), labels)
    got = labels.get()
    want = np.full((1, 480, 848), 65535, np.uint16)
    oracle_np.eval_forest(frame, forest_np, want)
    assert np.array_equal(got, want)
    valid = (frame != 0) & (frame != 65535)
    assert (got[~valid] == 65535).all() and got[valid].max() < 4 and 0.10 < valid.mean() < 0.20
    # single-tree entry point on the same data (every walk reaches a leaf in the full topology)
    tree = rdf.DecisionTree(10, 4)
    tree.tree_out_cu.set(forest_np[0])
    out = rdf.DeviceArray((1, 480, 848), np.uint16).fill(65535)
    ev.get_labels(tree, rdf.to_device(frame), out)
    assert np.array_equal(out.get(), want)


def test_filled_variant_needs_no_fill_first(rdf, host_runtime):
    """get_labels_forest_filled = the caller's fill of decision_tree.py:237-240 + get_labels_forest, whatever the label
    buffer held before; the reference-named method keeps the reference's signature and leaves untouched pixels alone."""
    g = np.load(GOLDEN)
    forest = rdf.DecisionForest.from_numpy(g["g1_forest"])
    depth = rdf.to_device(g["g1_depth"])
    for use_packed in (True, False):
        ev = rdf.DecisionTreeEvaluator(use_packed=use_packed)
        labels = rdf.DeviceArray(g["g1_labels"].shape, np.uint16).fill(1234)
        ev.get_labels_forest_filled(forest, depth, labels)
        assert np.array_equal(labels.get(), g["g1_labels"])
        plain = rdf.DeviceArray(g["g1_labels"].shape, np.uint16).fill(1234)
        ev.get_labels_forest(forest, depth, plain)
        untouched = g["g1_labels"] == 65535
        assert untouched.any() and np.all(plain.get()[untouched] == 1234)
        assert np.array_equal(plain.get()[~untouched], g["g1_labels"][~untouched])
