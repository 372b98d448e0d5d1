"""Rows H-K against the reference's OWN Python: tests/golden/callconv_v1.json was recorded by running
/root/reference/src/decision_tree.py on stand-in pycuda / OpenGL modules (tests/golden/make_callconv_golden.py, build
container only); here the same scenarios run through this repo's package on the host test double and must issue the same
sequence -- fills, forest evaluations in layer order, composite -- with the same buffers in the same roles and the same
scalars.  Plumbing, not arithmetic (the arithmetic is the oracle's job)."""
import json
import os

import numpy as np
import pytest

FIX = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "callconv_v1.json")))
SCEN = {s["scenario"]: s for s in FIX["scenarios"]}


def _ref_events(name):
    """The reference's recording in a launch-geometry-free form."""
    out = []
    for e in SCEN[name]["events"]:
        if e["op"] == "fill":
            out.append(("fill", e["buffer"], int(np.prod(e["shape"])), e["value"]))
        elif e["op"] == "launch" and e["kernel"] == "evaluate_image_using_forest":
            v = [a.get("scalar", a.get("buffer")) for a in e["args"]]
            T, n_img, dim_x, dim_y, C, D, bdx, depth, fcls, filt, forest, labels, r, s = v
            assert bdx == FIX["MAX_THREADS_PER_BLOCK"] // T and e["block"] == [bdx, T, 1] and e["shared"] == bdx * C * 4
            assert e["grid"] == [n_img * (dim_y // r) * (dim_x // r) // bdx + 1, 1, 1]
            assert [a["dtype"] for a in e["args"] if "scalar" in a] == ["int32"] * 8 + ["int32", "float32"]
            out.append(("forest", depth, forest, filt if fcls != -1 else None, fcls, labels, T, n_img, dim_x, dim_y, C, D, r, s))
        elif e["op"] == "launch" and e["kernel"] == "evaluate_image_using_tree":
            n_img, dim_x, dim_y, C, D, depth, tree, labels = [a.get("scalar", a.get("buffer")) for a in e["args"]]
            assert e["block"] == [1024, 1, 1] and e["grid"] == [n_img * dim_y * dim_x // 1024 + 1, 1, 1]
            out.append(("tree", depth, tree, labels, n_img, dim_x, dim_y, C, D))
        elif e["op"] == "launch" and e["kernel"] == "make_composite_labels_image":
            images, n, dim_x, dim_y, cond, out_img = [a.get("scalar", a.get("buffer")) for a in e["args"]]
            assert e["block"] == [32, 32, 1] and e["grid"] == [dim_x // 32 + 1, dim_y // 32 + 1, 1]
            out.append(("composite", images, n, dim_x, dim_y, cond, out_img))
    return out


def _my_events(trace, roles):
    """The package's calls into the C ABI (tests/fake_runtime.py's trace) in the same form."""
    def who(p):
        return None if p is None else roles.get(int(p), f"unknown@{int(p):#x}")
    out = []
    for name, a in trace:
        if name == "rdf_fill_u16":
            out.append(("fill", who(a[0]), int(a[1]), int(a[2])))
        elif name == "rdf_eval_forest":
            depth, n_img, dim_x, dim_y, forest, T, D, C, filt, fcls, labels, r, s, _ = a
            out.append(("forest", who(depth), who(forest), who(filt) if fcls != -1 else None, fcls, who(labels), T, n_img, dim_x, dim_y, C, D, r, float(s)))
        elif name == "rdf_eval_tree":
            depth, n_img, dim_x, dim_y, tree, D, C, labels, _ = a
            out.append(("tree", who(depth), who(tree), who(labels), n_img, dim_x, dim_y, C, D))
        elif name == "rdf_composite":
            images, n, dim_x, dim_y, cond, n_cond, out_img, bad, _ = a
            out.append(("composite", who(images), n, dim_x, dim_y, who(cond), who(out_img)))
    return out


def test_fixture_is_data_only():
    text = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "callconv_v1.json")).read()
    assert "def " not in text and "import " not in text and "__global__" not in text


def test_get_config_and_constructors(rdf, host_runtime):
    for key, want in FIX["get_config"].items():
        d, c = (int(v) for v in key.split(","))
        assert list(rdf.DecisionTree.get_config(d, c)) == want
    s = SCEN["constructors"]
    f = rdf.DecisionForest(4, 6, 4)
    assert list(f.forest_cu.shape) == s["forest_shape"] and f.forest_cu.dtype == np.float32
    assert {k: int(getattr(f, k)) for k in s["forest_attrs"]} == s["forest_attrs"]
    assert not f.forest_cu.get().any()                      # the reference fills its tables with 0.0f
    t = rdf.DecisionTree(5, 3)
    assert list(t.tree_out_cu.shape) == s["tree_shape"] and not t.tree_out_cu.get().any()


def test_flat_entry_points_issue_the_reference_calls(rdf, host_runtime):
    ev = rdf.DecisionTreeEvaluator(use_packed=False)
    lib = host_runtime.lib
    tree = rdf.DecisionTree(5, 3)
    depth, labels = rdf.DeviceArray((2, 48, 64), np.uint16).fill(1000), rdf.DeviceArray((2, 48, 64), np.uint16).fill(65535)
    del lib.trace[:]
    ev.get_labels(tree, depth, labels)
    roles = {depth.ptr: "depth", labels.ptr: "labels", tree.tree_out_cu.ptr: "tree"}
    assert _my_events(lib.trace, roles) == _ref_events("get_labels")

    f4 = rdf.DecisionForest(4, 6, 4)
    depth3, lab3 = rdf.DeviceArray((3, 48, 64), np.uint16).fill(1000), rdf.DeviceArray((3, 48, 64), np.uint16).fill(65535)
    del lib.trace[:]
    ev.get_labels_forest(f4, depth3, lab3)
    roles = {depth3.ptr: "depth", lab3.ptr: "labels", f4.forest_cu.ptr: "forest"}
    assert _my_events(lib.trace, roles) == _ref_events("get_labels_forest_r1")

    lab3h, filt = rdf.DeviceArray((3, 24, 32), np.uint16).fill(65535), rdf.DeviceArray((3, 24, 32), np.uint16).fill(2)
    del lib.trace[:]
    ev.get_labels_forest(f4, depth3, lab3h, labels_reduce=2, filter_images=filt, filter_images_class=2, scale_factor=0.5)
    roles = {depth3.ptr: "depth", lab3h.ptr: "labels", filt.ptr: "filter", f4.forest_cu.ptr: "forest"}
    assert _my_events(lib.trace, roles) == _ref_events("get_labels_forest_r2_filter")

    f3 = rdf.DecisionForest(3, 7, 7)
    del lib.trace[:]
    ev.get_labels_forest(f3, depth3, lab3, 1, None, None, 2.0)      # positional, as the reference's signature has it
    roles = {depth3.ptr: "depth", lab3.ptr: "labels", f3.forest_cu.ptr: "forest"}
    assert _my_events(lib.trace, roles) == _ref_events("get_labels_forest_three_trees_positional")


@pytest.mark.parametrize("label", ["layered_two_layers", "layered_three_layers"])
def test_layered_forest_issues_the_reference_sequence(rdf, host_runtime, tmp_path, label):
    run = SCEN[label + "_run"]
    init = SCEN[label + "_init"]
    for fn, shp in run["forest_shapes"].items():
        np.save(tmp_path / fn, np.zeros(shp, np.float32))
    cfg_path = tmp_path / (label + ".json")
    cfg_path.write_text(json.dumps(run["config"]))
    lf = rdf.LayeredDecisionForest.load(str(cfg_path), tuple(run["depth_dims"]), run["labels_reduce"])
    lf.fused = False                    # the reference's sequence, launch for launch (the one-call path is checked against it on the GPU)
    lf.eval.use_packed = False
    # what __init__ leaves behind
    assert list(lf.labels_dims) == init["labels_dims"] and list(lf.depth_dims) == init["depth_dims"]
    assert lf.num_models == init["num_models"] and lf.num_layered_classes == init["num_layered_classes"]
    assert [[m.num_trees, m.max_depth, m.num_classes, fm, fc] for m, fm, fc in lf.m] == init["layers"]
    sets = {e["buffer"]: e for e in init["events"] if e["op"] == "set"}
    roles = {rdf.device_ptr(b): f"layer{i}_labels" for i, b in enumerate(lf.label_images)}
    assert [roles[int(p)] for p in lf.labels_images_ptrs_cu.cu().get()] == sets["label_pointer_table"]["values_as_roles"]
    assert lf.labels_images_ptrs_cu.cu().dtype == np.int64
    assert lf.labels_conditions_cu.cu().get().reshape(-1).tolist() == sets["conditions"]["small_values"]
    assert lf.labels_conditions_cu.cu().dtype == np.int32 and list(lf.labels_conditions_cu.shape) == sets["conditions"]["shape"]
    assert list(lf.label_colors.shape) == sets["label_colors"]["shape"] and lf.label_colors.cu().dtype == np.uint8
    # run()
    depth_image = rdf.GpuBuffer(tuple(run["depth_dims"]), np.uint16)
    labels_image = rdf.GpuBuffer(lf.labels_dims, np.uint16)
    depth_image.cu().fill(1000)
    roles.update({rdf.device_ptr(depth_image): "depth_image", rdf.device_ptr(labels_image): "labels_image",
                  rdf.device_ptr(lf.labels_images_ptrs_cu): "label_pointer_table", rdf.device_ptr(lf.labels_conditions_cu): "conditions"})
    roles.update({m.forest_cu.ptr: f"layer{i}_forest" for i, (m, _, _) in enumerate(lf.m)})
    lib = host_runtime.lib
    del lib.trace[:]
    lf.run(depth_image, labels_image, run["scale_factor"])
    assert _my_events(lib.trace, roles) == _ref_events(label + "_run")


def test_proposals_are_the_references_own_bit_for_bit(rdf):
    """The reference's make_random_features (decision_tree.py:353-371) is pure numpy, so the fixture holds ITS output for a
    seeded global RNG: both generators of this package -- the draw-by-draw loop and the bulk one the trainer uses -- give the same
    float32 bits and leave the RNG where the reference leaves it."""
    dt = __import__("importlib").import_module("3d-beats_amd.decision_tree")
    fx = FIX["proposals"]
    want = np.array(fx["float32_bits"], dtype=np.uint32).reshape(fx["n"], 5)
    for gen in (dt.make_random_features_loop, dt.make_random_features):
        np.random.seed(fx["seed"])
        arr = np.zeros((fx["n"], 5), np.float32)
        gen(fx["n"], arr)
        assert np.array_equal(arr.view(np.uint32), want), gen.__name__
        assert float(np.random.random()).hex() == fx["next_random_after"], gen.__name__


def test_dataset_conversions_are_the_references_own(rdf, tmp_path):
    """DecisionTreeDatasetConfig's bookkeeping and colour <-> id conversions against the reference class's own outputs."""
    ds_mod = __import__("importlib").import_module("3d-beats_amd.dataset")
    fx = FIX["dataset"]
    (tmp_path / "config.json").write_text(json.dumps(fx["config"]))
    ds = ds_mod.DecisionTreeDatasetConfig(str(tmp_path))
    assert ds.num_classes() == fx["num_classes"] and list(ds.img_dims) == fx["img_dims"]
    assert ds.total_available_images == fx["total_available_images"]
    ids = np.array(fx["ids"], dtype=np.uint16)
    colors = ds.convert_ids_to_colors(ids)
    assert colors.dtype.name == fx["colors_dtype"] and np.array_equal(colors, np.array(fx["colors"], dtype=np.uint8))
    back = ds.convert_colors_to_ids(np.array(fx["colors"], dtype=np.uint8)[1])
    assert back.dtype.name == fx["back_dtype"] and back.tolist() == fx["ids_back_from_colors_of_image_1"]


def test_reference_app_constructor_runs_on_the_swapped_modules(rdf, host_runtime, tmp_path):
    """The constructor section of the reference's live app (run_live_layered.py:6-14, 20-58), import lines and calls as they
    stand there, on this package's modules (`install_reference_aliases`): the argument parser with py_nvcc_utils' two
    options, config_compiler, LayeredDecisionForest.load with (y, x) dims, PointsOps, the GpuBuffers -- and the recorded
    run() sequence afterwards.  (RealSense, the plane fit and the GL texture are out of scope: SURVEY section 2.)"""
    import argparse
    import sys
    run = SCEN["layered_two_layers_run"]
    for fn, shp in run["forest_shapes"].items():
        np.save(tmp_path / fn, np.zeros(shp, np.float32))
    cfg_path = tmp_path / "model_cfg.json"
    cfg_path.write_text(json.dumps(run["config"]))
    before = {n: sys.modules.get(n) for n in rdf._REFERENCE_MODULE_NAMES}
    try:
        assert rdf.install_reference_aliases() == list(rdf._REFERENCE_MODULE_NAMES)
        assert rdf.install_reference_aliases() == list(rdf._REFERENCE_MODULE_NAMES)        # (twice is harmless)
        import types
        sys.modules["util"] = types.ModuleType("util")          # somebody else's module of that name: refused unless forced
        with pytest.raises(ImportError, match="already imported"):
            rdf.install_reference_aliases()
        assert "util" in rdf.install_reference_aliases(force=True)
        ns = {}
        exec("from decision_tree import *\n"                      # run_live_layered.py:6
             "from cuda.points_ops import *\n"                    # :7
             "import cuda.py_nvcc_utils as py_nvcc_utils\n"       # :10
             "from engine.buffer import GpuBuffer\n"              # :14
             "from cuda.mean_shift import *\n"                    # 3d_bz.py
             "from util import MAX_UINT16\n", ns)
        py_nvcc_utils = ns["py_nvcc_utils"]
        parser = argparse.ArgumentParser(description='Train a classifier RDF for depth images')
        parser.add_argument('-cfg', nargs='?', required=True, type=str)
        parser.add_argument('--plane_num_iterations', nargs='?', required=False, type=int)
        py_nvcc_utils.add_args(parser)                                                  # :23
        args = parser.parse_args(["-cfg", str(cfg_path), "--fatbin_in", "fatbins/"])    # a command line written for the reference
        py_nvcc_utils.config_compiler(args)                                             # :27
        with pytest.raises(AssertionError):                                             # py_nvcc_utils.py:14
            py_nvcc_utils.config_compiler(parser.parse_args(["-cfg", "x", "--fatbin_in", "a", "--fatbin_out", "b"]))
        with pytest.raises(rdf.RdfError, match="prebuilt"):
            py_nvcc_utils.get_module("tree_eval")
        DIM_Y, DIM_X = run["depth_dims"]
        LABELS_REDUCE = run["labels_reduce"]
        layered_rdf = ns["LayeredDecisionForest"].load(args.cfg, (DIM_Y, DIM_X), LABELS_REDUCE)        # :54
        points_ops = ns["PointsOps"]()                                                                  # :55
        GpuBuffer = ns["GpuBuffer"]
        depth_image = GpuBuffer((1, DIM_Y, DIM_X), np.uint16)                                           # :59
        labels_image = GpuBuffer((1, DIM_Y // LABELS_REDUCE, DIM_X // LABELS_REDUCE), dtype=np.uint16)  # :62
        assert ns["MAX_UINT16"] == 65535 and points_ops is not None
        assert layered_rdf.__class__ is rdf.LayeredDecisionForest and GpuBuffer is rdf.GpuBuffer
        # tick(): upload, 0 -> 65535, run (run_live_layered.py:117-126)
        frame = np.full((1, DIM_Y, DIM_X), 1000, np.uint16)
        frame[0, :4] = 0
        depth_image.cu().set(frame)
        points_ops.convert_0s_to_maxuint(np.int32(DIM_X * DIM_Y), depth_image.cu(), grid=((DIM_X * DIM_Y) // 1024 + 1, 1, 1), block=(1024, 1, 1))
        assert (depth_image.cu().get()[0, :4] == 65535).all()
        layered_rdf.fused = False
        layered_rdf.eval.use_packed = False
        lib = host_runtime.lib
        del lib.trace[:]
        layered_rdf.run(depth_image, labels_image, float(run["scale_factor"]))
        roles = {rdf.device_ptr(b): f"layer{i}_labels" for i, b in enumerate(layered_rdf.label_images)}
        roles.update({rdf.device_ptr(depth_image): "depth_image", rdf.device_ptr(labels_image): "labels_image",
                      rdf.device_ptr(layered_rdf.labels_images_ptrs_cu): "label_pointer_table",
                      rdf.device_ptr(layered_rdf.labels_conditions_cu): "conditions"})
        roles.update({m.forest_cu.ptr: f"layer{i}_forest" for i, (m, _, _) in enumerate(layered_rdf.m)})
        assert _my_events(lib.trace, roles) == _ref_events("layered_two_layers_run")
    finally:
        for n, m in before.items():
            if m is None:
                sys.modules.pop(n, None)
            else:
                sys.modules[n] = m
