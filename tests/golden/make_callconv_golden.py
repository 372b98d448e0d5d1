#!/usr/bin/env python3
"""Captures the CALL CONVENTION of the reference's own host code (rows H-K of SURVEY 8a) into tests/golden/callconv_v1.json.

Runs in the build container only (it imports /root/reference/src/decision_tree.py, which never travels to the GPU box):
stand-ins for the modules the reference imports but this image lacks -- pycuda.{compiler,driver,gpuarray,gl}, nvcomp,
OpenGL.GL -- are put into sys.modules; they allocate nothing and compute nothing, they RECORD: every GPUArray.fill / .set,
and every kernel launch (function name, typed scalar arguments, which buffer every pointer argument is, grid, block, shared
bytes).  The reference's Python then runs as it is -- DecisionTree.get_config, DecisionTreeEvaluator.get_labels /
get_labels_forest (r = 1 and 2, with and without a filter), LayeredDecisionForest.__init__ + run (two and three layers) --
and the recorded sequences are the fixture.  tests/test_callconv.py replays the same scenarios through this repo's package
on tests/fake_runtime.py and asserts the same sequence, wiring and scalars.

This pins PLUMBING (fill order, layer order, filter wiring, (y, x) dims, argument order and types, launch geometry), not
the kernels' arithmetic: no kernel runs.  Two things in the fixture ARE outputs of the reference, because they are pure numpy
and run as they are: the proposal generator (decision_tree.py:353-371: 40 proposals from a seeded global RNG, bit patterns and
the RNG's position afterwards) and the dataset class's colour <-> id conversions (decision_tree.py:46-122).  The fixture holds data only -- names, shapes and numbers -- no reference source text.

    python3 tests/golden/make_callconv_golden.py        # rewrites tests/golden/callconv_v1.json
"""
import json
import os
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
EVENTS = []          # the recording, in program order
ROLES = {}           # fake device address -> role name given by the scenario
_next_addr = [0x1000]


def _name(addr):
    return ROLES.get(addr, f"buffer@{addr:#x}")


class GPUArray:
    """Stand-in for pycuda.gpuarray.GPUArray: a shape, a dtype and a fake address; views share the address."""

    def __init__(self, shape, dtype, gpudata=None, **_):
        self.shape = tuple(int(v) for v in (shape if isinstance(shape, (tuple, list)) else (shape,)))
        self.dtype = np.dtype(dtype)
        if gpudata is None:
            gpudata = _next_addr[0]
            _next_addr[0] += 0x1000
        self.gpudata = int(gpudata)
        self.ptr = self.gpudata

    @property
    def __cuda_array_interface__(self):
        return {"data": (self.gpudata, False), "shape": self.shape, "typestr": self.dtype.str, "version": 2}

    def fill(self, value):
        EVENTS.append({"op": "fill", "buffer": self.gpudata, "shape": list(self.shape), "dtype": self.dtype.name,
                       "value": np.asarray(value).item(), "value_dtype": np.asarray(value).dtype.name})
        return self

    def set(self, arr):
        arr = np.asarray(arr)
        EVENTS.append({"op": "set", "buffer": self.gpudata, "shape": list(arr.shape), "dtype": arr.dtype.name,
                       "small_values": arr.reshape(-1).tolist() if arr.size <= 64 and arr.dtype.kind in "iu" else None})

    def reshape(self, *shape):
        shape = shape[0] if len(shape) == 1 and isinstance(shape[0], (tuple, list)) else shape
        assert int(np.prod(shape)) == int(np.prod(self.shape))
        return GPUArray(shape, self.dtype, gpudata=self.gpudata)

    def __bool__(self):      # the reference tests `if filter_images` (decision_tree.py:313): an array it was handed counts as true
        return True


class _Function:
    def __init__(self, name):
        self.name = name

    def __call__(self, *args, grid=None, block=None, shared=0, **kw):
        rec = []
        for a in args:
            if isinstance(a, GPUArray):
                rec.append({"buffer": a.gpudata, "shape": list(a.shape), "dtype": a.dtype.name})
            elif isinstance(a, np.generic):
                rec.append({"scalar": a.item(), "dtype": a.dtype.name})
            else:
                rec.append({"python": repr(a)})
        EVENTS.append({"op": "launch", "kernel": self.name, "args": rec, "grid": list(grid), "block": list(block),
                       "shared": int(shared)})


class _SourceModule:
    def __init__(self, source, **kw):
        self.kw = {k: v for k, v in kw.items() if k != "include_dirs"}

    def get_function(self, name):
        return _Function(name)


def install_stand_ins():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class _Mapping:
        def __init__(self, addr):
            self.addr = addr

        def device_ptr_and_size(self):
            return (self.addr, 0)

        def unmap(self):
            pass

    class RegisteredBuffer:
        def __init__(self, gl_id):
            self.addr = _next_addr[0]
            _next_addr[0] += 0x1000

        def map(self):
            return _Mapping(self.addr)

    pycuda = mod("pycuda")
    pycuda.compiler = mod("pycuda.compiler", SourceModule=_SourceModule, FATBIN_IN_DIR=None)
    pycuda.driver = mod("pycuda.driver", pagelocked_zeros=lambda shape, dtype: np.zeros(shape, dtype))
    pycuda.gpuarray = mod("pycuda.gpuarray", GPUArray=GPUArray)
    pycuda.gl = mod("pycuda.gl", RegisteredBuffer=RegisteredBuffer)
    mod("nvcomp")
    gl_ids = [0]

    def glGenBuffers(n):
        gl_ids[0] += 1
        return gl_ids[0]

    ogl = mod("OpenGL")
    ogl.GL = mod("OpenGL.GL", GL_DYNAMIC_DRAW=0x88E8, GL_ARRAY_BUFFER=0x8892, glGenBuffers=glGenBuffers,
                 glBindBuffer=lambda *a: None, glBufferData=lambda *a: None)
    if "PIL" not in sys.modules:
        try:
            import PIL.Image  # noqa: F401
        except Exception:
            pil = mod("PIL")
            pil.Image = mod("PIL.Image")
    # aliases numpy has since removed, which the reference still uses (decision_tree.py:275, :446)
    for alias, typ in (("int", int), ("float", float), ("bool", bool)):
        if not hasattr(np, alias):
            setattr(np, alias, typ)


def take(label):
    """The events recorded since the last take, with addresses replaced by role names."""
    out = []
    for e in EVENTS:
        e = json.loads(json.dumps(e))
        if e["op"] == "set" and e["dtype"] == "int64" and e["small_values"] and all(v in ROLES for v in e["small_values"]):
            e["values_as_roles"] = [_name(v) for v in e.pop("small_values")]      # a table of device addresses
        if "buffer" in e:
            e["buffer"] = _name(e["buffer"])
        for a in e.get("args", []):
            if "buffer" in a:
                a["buffer"] = _name(a["buffer"])
        out.append(e)
    del EVENTS[:]
    return {"scenario": label, "events": out}


def role(obj, name):
    arr = obj.cu() if hasattr(obj, "cu") else obj
    ROLES[arr.gpudata] = name


def main():
    install_stand_ins()
    os.chdir(REF)                                    # py_nvcc_utils.get_module opens ./src/cuda/<n>.cu
    sys.path.insert(0, os.path.join(REF, "src"))
    import decision_tree as ref                      # the reference's own module
    from engine.buffer import GpuBuffer

    fixture = {"what": "call convention of /root/reference/src/decision_tree.py recorded through stand-in pycuda / OpenGL / nvcomp "
                       "modules (tests/golden/make_callconv_golden.py); plumbing only, no kernel ran",
               "get_config": {f"{d},{c}": list(ref.DecisionTree.get_config(d, c)) for d, c in [(20, 4), (10, 3), (1, 1), (22, 4), (18, 7)]},
               "MAX_THREADS_PER_BLOCK": int(ref.MAX_THREADS_PER_BLOCK), "scenarios": []}

    # ---- DecisionTree / DecisionForest constructors ----
    t = ref.DecisionTree(5, 3)
    role(t.tree_out_cu, "tree")
    f4 = ref.DecisionForest(4, 6, 4)
    role(f4.forest_cu, "forest")
    fixture["scenarios"].append(dict(take("constructors"), tree_shape=list(t.tree_out_cu.shape), forest_shape=list(f4.forest_cu.shape),
                                     forest_attrs={k: int(getattr(f4, k)) for k in ("num_trees", "max_depth", "num_classes", "TOTAL_TREE_NODES",
                                                                                   "MAX_LEAF_NODES", "TREE_NODE_ELS")}))

    ev = ref.DecisionTreeEvaluator()
    role(ev.cu_empty_ptr, "evaluator_dummy")
    fixture["scenarios"].append(take("evaluator_init"))

    # ---- get_labels (single tree) ----
    depth = GPUArray((2, 48, 64), np.uint16)
    labels = GPUArray((2, 48, 64), np.uint16)
    role(depth, "depth"); role(labels, "labels")
    ev.get_labels(t, depth, labels)
    fixture["scenarios"].append(take("get_labels"))

    # ---- get_labels_forest: r = 1 without filter, r = 2 with filter, three trees ----
    depth3 = GPUArray((3, 48, 64), np.uint16)
    lab3 = GPUArray((3, 48, 64), np.uint16)
    role(depth3, "depth"); role(lab3, "labels")
    ev.get_labels_forest(f4, depth3, lab3)
    fixture["scenarios"].append(take("get_labels_forest_r1"))
    lab3h = GPUArray((3, 24, 32), np.uint16)
    filt = GPUArray((3, 24, 32), np.uint16)
    role(lab3h, "labels"); role(filt, "filter")
    ev.get_labels_forest(f4, depth3, lab3h, labels_reduce=2, filter_images=filt, filter_images_class=2, scale_factor=0.5)
    fixture["scenarios"].append(take("get_labels_forest_r2_filter"))
    f3 = ref.DecisionForest(3, 7, 7)
    role(f3.forest_cu, "forest")
    del EVENTS[:]
    ev.get_labels_forest(f3, depth3, lab3, 1, None, None, 2.0)
    fixture["scenarios"].append(take("get_labels_forest_three_trees_positional"))

    # ---- LayeredDecisionForest: load + run ----
    tmp = tempfile.mkdtemp(prefix="callconv_")
    shapes = {"hand.npy": (4, 2 ** 6 - 1, 7 + 2 * 3), "fingers.npy": (3, 2 ** 5 - 1, 7 + 2 * 4), "tips.npy": (2, 2 ** 4 - 1, 7 + 2 * 2)}
    for fn, shp in shapes.items():
        np.save(os.path.join(tmp, fn), np.zeros(shp, np.float32))
    for label, cfg, dims, r, s in [
        ("layered_two_layers", {"layers": [{"model": "hand.npy"}, {"model": "fingers.npy", "filter_model": 0, "filter_model_class": 3}],
                                "conditions": [[0, 1], [0, 2], [1, 3], [0, 3], [0, 4]], "label_colors": [[1, 2, 3, 255]] * 4}, (48, 64), 2, 0.5),
        ("layered_three_layers", {"layers": [{"model": "hand.npy"}, {"model": "fingers.npy", "filter_model": 0, "filter_model_class": 3},
                                             {"model": "tips.npy"}],
                                  "conditions": [[0, 1], [0, 2], [1, 3], [0, 3], [0, 4]], "label_colors": [[9, 9, 9, 255]] * 4}, (30, 40), 1, 1.0)]:
        path = os.path.join(tmp, label + ".json")
        json.dump(cfg, open(path, "w"))
        del EVENTS[:]
        lf = ref.LayeredDecisionForest.load(path, dims, r)
        role(lf.eval.cu_empty_ptr, "evaluator_dummy")
        for i, b in enumerate(lf.label_images):
            role(b, f"layer{i}_labels")
        for i, (m, _, _) in enumerate(lf.m):
            role(m.forest_cu, f"layer{i}_forest")
        role(lf.labels_images_ptrs_cu, "label_pointer_table")
        role(lf.labels_conditions_cu, "conditions")
        role(lf.label_colors, "label_colors")
        init = take(label + "_init")
        init.update({"labels_dims": list(lf.labels_dims), "depth_dims": list(lf.depth_dims), "num_models": int(lf.num_models),
                     "num_layered_classes": int(lf.num_layered_classes),
                     "layers": [[int(m.num_trees), int(m.max_depth), int(m.num_classes), fm, fc] for m, fm, fc in lf.m]})
        fixture["scenarios"].append(init)
        depth_image = GpuBuffer(dims, np.uint16)
        labels_image = GpuBuffer(lf.labels_dims, np.uint16)
        role(depth_image, "depth_image"); role(labels_image, "labels_image")
        lf.run(depth_image, labels_image, s)
        fixture["scenarios"].append(dict(take(label + "_run"), config=cfg, depth_dims=list(dims), labels_reduce=r, scale_factor=s,
                                         forest_shapes={k: list(v) for k, v in shapes.items()}))
    # ---- the reference's pure-numpy helpers, run as they are: these fixtures ARE outputs of the reference ----
    # proposals (decision_tree.py:353-371): values bit for bit, and where the global RNG stands afterwards
    np.random.seed(20211003)
    arr = np.zeros((40, 5), np.float32)
    ref.make_random_features(40, arr)
    fixture["proposals"] = {"seed": 20211003, "n": 40, "float32_bits": arr.view(np.uint32).reshape(-1).tolist(),
                            "next_random_after": float(np.random.random()).hex()}
    # dataset bookkeeping and the colour <-> id conversions (decision_tree.py:46-122) on a tiny config (num_images = 0: no image blocks)
    ddir = os.path.join(tmp, "ds") + os.sep
    os.makedirs(ddir)
    ds_cfg = {"img_dims": [6, 4], "num_images": 3, "id_to_color": {"1": [255, 0, 0, 255], "2": [0, 255, 0, 255], "3": [10, 20, 30, 255]}}
    json.dump(ds_cfg, open(ddir + "config.json", "w"))
    ds = ref.DecisionTreeDatasetConfig(ddir)
    rng = np.random.default_rng(5)
    ids = rng.integers(0, 4, size=(2, 4, 6)).astype(np.uint16)
    colors = ds.convert_ids_to_colors(ids)
    back = ds.convert_colors_to_ids(colors[1])
    fixture["dataset"] = {"config": ds_cfg, "num_classes": int(ds.num_classes()), "img_dims": list(ds.img_dims),
                          "total_available_images": int(ds.total_available_images), "ids": ids.tolist(), "colors": colors.tolist(),
                          "ids_back_from_colors_of_image_1": back.tolist(), "back_dtype": back.dtype.name, "colors_dtype": colors.dtype.name}
    out = os.path.join(HERE, "callconv_v1.json")
    json.dump(fixture, open(out, "w"), indent=1)
    print(f"wrote {out}: {len(fixture['scenarios'])} scenarios, {sum(len(s_['events']) for s_ in fixture['scenarios'])} events")


if __name__ == "__main__":
    main()
