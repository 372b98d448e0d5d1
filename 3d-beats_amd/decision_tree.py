"""Host side of the RDF inference path: the reference's `decision_tree.py` surface for inference.

Mirrors, name for name, the classes callers of /root/reference/src/decision_tree.py use on the
inference path (run_live_layered.py:54-58,126; 3d_bz.py:76-110,389-437; run_live.py:123-124;
test_on_saved_model.py:54-56):

    DecisionTree(max_depth, num_classes), DecisionTree.get_config           decision_tree.py:124-144
    DecisionForest(num_trees, max_depth, num_classes), DecisionForest.load  decision_tree.py:146-168
    LayeredDecisionForest.load / .run and its public attributes             decision_tree.py:171-264
    DecisionTreeEvaluator.get_labels / .get_labels_forest /
        .make_composite_labels_image                                        decision_tree.py:267-347

Kernels are reached through the C ABI of librdf_hip.so (include/rdf_hip.h); launch geometry is the
library's business.  The trainer half of the reference module (decision_tree.py:353-600) is further down:
make_random_features and DecisionTreeTrainer; the dataset reader lives in dataset.py.
"""
import json
import logging
import os
import time

import numpy as np

from . import _lib
from .device import DeviceArray, device_ptr, get_runtime
from .engine.buffer import GpuBuffer
from .util import MAX_UINT16  # noqa: F401  (re-exported like the reference module)

_PIX_LIMIT = (1 << 31) - 1  # one C-ABI call addresses < 2^31 depth pixels (RDF_ERR_TOO_LARGE)
_log = logging.getLogger("rdf_hip")


class DecisionTree:
    def __init__(self, max_depth, num_classes):
        self.max_depth = max_depth
        self.num_classes = num_classes
        self.TOTAL_TREE_NODES, self.MAX_LEAF_NODES, self.TREE_NODE_ELS = DecisionTree.get_config(max_depth, num_classes)
        # tightly packed level-order binary tree, all-zero = untrained
        self.tree_out_cu = DeviceArray((self.TOTAL_TREE_NODES, self.TREE_NODE_ELS), np.float32)
        self.tree_out_cu.fill(np.float32(0.))

    @staticmethod
    def get_config(max_depth, num_classes):
        total_tree_nodes = (2 ** max_depth) - 1     # nodes of a complete tree
        max_leaf_nodes = 2 ** max_depth             # children of the deepest level
        tree_node_els = 7 + (num_classes * 2)       # ux,uy,vx,vy,thresh,l_next,r_next,l_pdf[C],r_pdf[C]
        return (total_tree_nodes, max_leaf_nodes, tree_node_els)


class DecisionForest:
    @staticmethod
    def load(model_filename):
        forest_cpu = np.load(model_filename)
        assert forest_cpu.ndim == 3, "forest .npy must be [trees, 2^D-1, 7+2C]"
        num_trees = forest_cpu.shape[0]
        tree_depth = int(np.log2(forest_cpu.shape[1] + 1))
        num_classes = (forest_cpu.shape[2] - 7) // 2
        f = DecisionForest(num_trees, tree_depth, num_classes)
        assert forest_cpu.shape == f.forest_cu.shape, \
            f"{model_filename}: shape {forest_cpu.shape} is not a complete depth-{tree_depth} forest"
        f.forest_cu.set(np.ascontiguousarray(forest_cpu, dtype=np.float32))
        return f

    @staticmethod
    def from_numpy(forest_cpu):
        """Convenience (not in the reference): build from an in-memory [T, 2^D-1, 7+2C] float32 array."""
        forest_cpu = np.ascontiguousarray(forest_cpu, dtype=np.float32)
        T, n, e = forest_cpu.shape
        f = DecisionForest(T, int(np.log2(n + 1)), (e - 7) // 2)
        assert forest_cpu.shape == f.forest_cu.shape
        f.forest_cu.set(forest_cpu)
        return f

    def __init__(self, num_trees, max_depth, num_classes):
        self.num_trees = num_trees
        self.max_depth = max_depth
        self.num_classes = num_classes
        self.TOTAL_TREE_NODES, self.MAX_LEAF_NODES, self.TREE_NODE_ELS = DecisionTree.get_config(max_depth, num_classes)
        self.forest_cu = DeviceArray((self.num_trees, self.TOTAL_TREE_NODES, self.TREE_NODE_ELS), np.float32)
        self.forest_cu.fill(np.float32(0.))
        self._packed = {}  # scale_factor -> (forest_cu identity, version, DeviceArray)

    def packed(self, scale_factor=1.):
        """Packed tables (16-byte hot records, PDF rows, last-level records, deep blocks) for `scale_factor`, rebuilt when forest_cu
        has been written since.

        The reference has no such step (its load is the upload at decision_tree.py:148-158); this is
        the load-time repack described in include/rdf_hip.h.  Returns None for an empty forest."""
        s = float(np.float32(scale_factor))
        key = (id(self.forest_cu), self.forest_cu.version)
        hit = self._packed.get(s)
        if hit is not None and hit[0] == key:
            return hit[1]
        rt = get_runtime()
        lib = rt.lib
        nbytes = int(lib.rdf_forest_packed_bytes(int(self.num_trees), int(self.max_depth), int(self.num_classes)))
        if nbytes == 0:
            return None
        if hit is not None and hit[1].nbytes != nbytes:
            self._forget(hit[1])
        buf = hit[1] if (hit is not None and hit[1].nbytes == nbytes) else DeviceArray((nbytes,), np.uint8)
        _lib.check(lib, lib.rdf_forest_pack(device_ptr(self.forest_cu), int(self.num_trees), int(self.max_depth),
                                            int(self.num_classes), s, buf.ptr, rt.stream()), "rdf_forest_pack")
        self._packed[s] = (key, buf)
        self.__dict__.setdefault("_tuned", {}).pop(s, None)      # a new table: its deep-level choice is made again
        return buf

    @staticmethod
    def _forget(buf):
        """The library keeps what it knows about a packed table per (device, address): told before the memory goes away, so that
        another table that later lands on the same address is read afresh (rdf_forest_forget)."""
        try:
            from . import device
            rt = device._runtime            # (never CREATE a runtime for this: a finalizer may run at interpreter exit)
            if rt is not None and hasattr(rt.lib, "rdf_forest_forget"):
                rt.lib.rdf_forest_forget(buf.ptr)
        except Exception:       # noqa: BLE001 -- (interpreter shutdown, or a runtime that is already gone)
            pass

    def __del__(self):
        for _, buf in list(getattr(self, "_packed", {}).values()):
            self._forget(buf)

    def packed_bytes(self, scale_factor=1.):
        """The packed table for `scale_factor` as host bytes (numpy uint8), deep-level choice included: pack and tune once where
        the model is built, ship the table, `adopt_packed` it where the model runs -- no repack, no tuning there.  Not in the
        reference."""
        buf = self.packed(scale_factor)
        return None if buf is None else buf.get().copy()

    def adopt_packed(self, table_bytes, scale_factor=1.):
        """Takes a table `packed_bytes` produced for THIS forest (same trees / depth / classes, same forest_cu contents, same
        scale) instead of packing: the table is uploaded and the library reads its info block AT ONCE (`_verify_table`): bytes
        that this library version's rdf_forest_pack did not write for this shape, or a table packed for another scale than
        `scale_factor`, raise ValueError here -- not at some later evaluation, where a first look at the table cannot be
        recorded into a hipGraph, and not never (the library evaluates with the table's own scale, so a wrong cache key would
        give another scale's labels silently).  The deep-level choice is found in the table."""
        s = float(np.float32(scale_factor))
        rt = get_runtime()
        lib = rt.lib
        nbytes = int(lib.rdf_forest_packed_bytes(int(self.num_trees), int(self.max_depth), int(self.num_classes)))
        table_bytes = np.ascontiguousarray(table_bytes, dtype=np.uint8).reshape(-1)
        if table_bytes.size != nbytes:
            raise ValueError(f"adopt_packed: a packed table of {table_bytes.size} bytes; this forest's "
                             f"(T{self.num_trees}/D{self.max_depth}/C{self.num_classes}) is {nbytes}")
        hit = self._packed.get(s)
        buf = hit[1] if (hit is not None and hit[1].nbytes == nbytes) else DeviceArray((nbytes,), np.uint8)
        self._forget(buf)               # whatever the library knew about this address
        self._packed.pop(s, None)       # (nothing half-adopted stays behind a failing check)
        buf.set(table_bytes)
        self._verify_table(buf, s, "adopt_packed")
        self._packed[s] = ((id(self.forest_cu), self.forest_cu.version), buf)
        self.__dict__.setdefault("_tuned", {}).pop(s, None)
        return buf

    def _verify_table(self, buf, s, who):
        """The library's first look at a table that arrived by other means than rdf_forest_pack (upload, broadcast): magic,
        shape and generation are checked by rdf_forest_info, the scale here.  Synchronous (a 128-byte read-back); afterwards the
        table is known, so a later evaluation can be captured into a hipGraph."""
        import ctypes
        rt = get_runtime()
        lib = rt.lib
        if not hasattr(lib, "rdf_forest_info"):         # (the host test double)
            return
        scale = ctypes.c_float(0.)
        rc = lib.rdf_forest_info(buf.ptr, int(self.num_trees), int(self.max_depth), int(self.num_classes), rt.stream(), None, None,
                                 ctypes.byref(scale))
        if rc != 0:
            self._forget(buf)
            msg = lib.rdf_error_string(int(rc))
            raise ValueError(f"{who}: these bytes are not a packed table of a T{self.num_trees}/D{self.max_depth}/C{self.num_classes} "
                             f"forest written by this library version's rdf_forest_pack "
                             f"({msg.decode() if isinstance(msg, bytes) else msg}, code {rc})")
        if np.float32(scale.value) != np.float32(s):
            self._forget(buf)
            raise ValueError(f"{who}: the table was packed for scale_factor {scale.value!r}, not {s!r}")

    def deep_from(self, scale_factor=1.):
        """The deep-level choice the packed table of `scale_factor` carries: a level, 0 (heap-order records), or None when
        nothing was chosen for it (rdf_forest_info; reads the table's info block if the library has not seen the table yet)."""
        import ctypes
        hit = self._packed.get(float(np.float32(scale_factor)))
        rt = get_runtime()
        lib = rt.lib
        if hit is None or not hasattr(lib, "rdf_forest_info"):
            return None
        level = ctypes.c_int(-1)
        _lib.check(lib, lib.rdf_forest_info(hit[1].ptr, int(self.num_trees), int(self.max_depth), int(self.num_classes), rt.stream(),
                                            ctypes.byref(level), None, None), "rdf_forest_info")
        return None if level.value < 0 else int(level.value)

    def tune(self, depth_images_in, labels_reduce=1, scale_factor=1.):
        """Chooses, by measurement on `depth_images_in` (device array [N, H, W] of representative frames), which table
        serves this packed forest's deep levels -- heap-order records or the deep blocks, and from which level
        (rdf_forest_tune, include/rdf_hip.h) -- and writes the choice into the packed table of `scale_factor`, where every
        later evaluation (and `packed_bytes` / `adopt_packed`) finds it.  The documented flow: `DecisionForest.load(...)`,
        then ONE `tune(sample)` at load time; a table nobody tuned is walked from the heap-order records.  Labels do
        not depend on the choice.  Not in the reference.  Returns {"deep_from": level or 0, "tried": {level: ms}}."""
        import ctypes
        packed = self.packed(scale_factor)
        if packed is None:
            return {"deep_from": 0, "tried": {}}
        rt = get_runtime()
        lib = rt.lib
        n, h, w = (int(v) for v in depth_images_in.shape)
        scratch = DeviceArray((n, h // labels_reduce, w // labels_reduce), np.uint16)
        chosen, tried = ctypes.c_int(0), ctypes.c_int(0)
        levels, ms = (ctypes.c_int * 12)(), (ctypes.c_float * 12)()
        _lib.check(lib, lib.rdf_forest_tune(device_ptr(depth_images_in), n, w, h, packed.ptr, device_ptr(self.forest_cu),
                                            int(self.num_trees), int(self.max_depth), int(self.num_classes), scratch.ptr,
                                            int(labels_reduce), rt.stream(), ctypes.byref(chosen), ctypes.byref(tried),
                                            levels, ms), "rdf_forest_tune")
        res = {"deep_from": int(chosen.value), "tried": {int(levels[i]): round(float(ms[i]), 4) for i in range(tried.value)}}
        self.__dict__.setdefault("_tuned", {})[float(np.float32(scale_factor))] = res      # (an evaluator's auto-tune then leaves it alone)
        return res


class LayeredDecisionForest:
    """Stack of forests evaluated in order, later layers optionally filtered on an earlier layer's
    labels, merged by a `conditions` table (decision_tree.py:171-264).  Owns its device buffers."""

    @staticmethod
    def load(config_filename, depth_dims, labels_reduce=1):
        with open(config_filename) as fh:
            cfg = json.loads(fh.read())
        # models are loaded 1-by-1 from paths relative to the config file
        cfg['root'] = os.path.dirname(os.path.abspath(config_filename))
        return LayeredDecisionForest(cfg, depth_dims, labels_reduce)

    def __init__(self, cfg, depth_dims, labels_reduce, fused=True):
        self.eval = DecisionTreeEvaluator()
        self.fused = fused          # run() through rdf_layered_run (one call) or step by step like the reference
        self._fused_args = None

        self.depth_dims = tuple(depth_dims)  # y,x !!
        self.labels_reduce = labels_reduce
        self.labels_dims = (depth_dims[0] // labels_reduce, depth_dims[1] // labels_reduce)

        self.m = []
        for l in cfg['layers']:
            m = l['model']
            if not isinstance(m, DecisionForest):
                m = DecisionForest.load(os.path.join(cfg.get('root', ''), m))
            # the reference's test is effectively `'filter_model' in l` (decision_tree.py:192)
            if 'filter_model' in l:
                filter_model = l['filter_model']
                filter_model_class = l['filter_model_class']
            else:
                filter_model = None
                filter_model_class = None
            self.m.append((m, filter_model, filter_model_class))

        self.num_models = len(self.m)
        for i, (_, fm, _) in enumerate(self.m):
            assert fm is None or 0 <= fm < self.num_models, f"layer {i}: filter_model {fm} out of range"
            # A layer that filters on itself or on a LATER layer reads labels that this frame has not produced yet.  The
            # reference fills every label buffer with 65535 first (decision_tree.py:237-240), so such a layer sees only
            # 65535; the fused call folds the fills into the kernels and would read the previous frame's labels instead:
            # such stacks take the step-by-step path, which is the reference's sequence launch for launch.
            if fm is not None and fm >= i:
                self.fused = False

        self.label_images = [GpuBuffer(self.labels_dims, dtype=np.uint16) for _ in range(self.num_models)]

        # device table of the per-layer label image addresses (decision_tree.py:205-207)
        self.labels_images_ptrs_cu = GpuBuffer((self.num_models,), dtype=np.int64)
        label_images_ptrs = np.array([device_ptr(i) for i in self.label_images], dtype=np.int64)
        self.labels_images_ptrs_cu.cu().set(label_images_ptrs)

        # conditions rows: (0, PIXEL_ID) or (1, NEXT_IMG_CONDITION_OFFSET); see decision_tree.py:209-220
        labels_conditions = np.array(cfg['conditions'], dtype=np.int32)
        assert labels_conditions.ndim == 2 and labels_conditions.shape[1] == 2
        self._cfg_host = {'conditions': labels_conditions.tolist(), 'label_colors': np.array(cfg['label_colors']).tolist()}
        self.labels_conditions_cu = GpuBuffer(labels_conditions.shape, dtype=np.int32)
        self.labels_conditions_cu.cu().set(labels_conditions)
        self.num_layered_classes = int(max([c[1] for c in filter(lambda c: c[0] == 0, labels_conditions)]))

        label_colors = np.array(cfg['label_colors'], dtype=np.uint8)
        assert label_colors.shape == (self.num_layered_classes, 4)
        self.label_colors = GpuBuffer(label_colors.shape, dtype=np.uint8)
        self.label_colors.cu().set(label_colors)

    def sibling(self):
        """A second stack over the SAME forests (device tables and packed tables are shared), conditions and colours, with
        its own per-layer label buffers, pointer table and evaluator.  run() writes `label_images` and the evaluator's
        error counter, so two runs that may be in flight together -- the two hands of a frame on two streams, two captured
        graphs -- must not go through one LayeredDecisionForest object; give each its own sibling.  (The reference app
        shares one object between the hands because it runs them one after the other on one stream, 3d_bz.py:281-284.)"""
        layers = []
        for m, fm, fc in self.m:
            l = {'model': m}
            if fm is not None:
                l['filter_model'], l['filter_model_class'] = fm, fc
            layers.append(l)
        cfg = {'layers': layers, 'conditions': self._cfg_host['conditions'], 'label_colors': self._cfg_host['label_colors']}
        other = LayeredDecisionForest(cfg, self.depth_dims, self.labels_reduce, fused=self.fused)
        other.eval.use_packed = self.eval.use_packed
        return other

    def run(self, depth_image, labels_image, scale_factor=1.):
        """Per-frame entry of the live apps (run_live_layered.py:126, 3d_bz.py:389-437): every label buffer ends
        up 65535 where nothing was classified, layer i holds forest i's labels, labels_image the composite."""
        if self.fused:
            return self._run_fused(depth_image, labels_image, scale_factor)

        labels_image.cu().fill(MAX_UINT16)
        for buf in self.label_images:
            buf.cu().fill(MAX_UINT16)

        # one image per call: add the leading image axis
        depth_3d = depth_image.cu().reshape((1,) + self.depth_dims)
        label_shape = (1,) + self.labels_dims
        for (forest, filter_model, filter_model_class), out_buf in zip(self.m, self.label_images):
            filt = self.label_images[filter_model].cu().reshape(label_shape) if filter_model is not None else None
            self.eval.get_labels_forest(forest, depth_3d, out_buf.cu().reshape(label_shape),
                                        labels_reduce=self.labels_reduce, filter_images=filt,
                                        filter_images_class=filter_model_class, scale_factor=scale_factor)

        self.eval.make_composite_labels_image(self.labels_images_ptrs_cu.cu(), self.labels_dims[1], self.labels_dims[0],
                                              self.labels_conditions_cu.cu(), labels_image.cu().reshape(label_shape))

    def run_hand(self, depth_image, labels_image, scale_factor=1., flip_x=False, color_image=None):
        """run() plus what the app does with one hand's composite (3d_bz.py:440-456) inside the same call: the composite
        is stored mirrored in x when `flip_x` (the left hand's labels go back to camera orientation) and `color_image`
        (RGBA bytes, labels_dims + (4,)) receives label_colors[label - 1] wherever a label in 1..num_layered_classes is
        written, as make_rgba_from_labels would.  The per-layer label images stay unflipped.  Stacks that cannot take the
        fused call (a layer filtering on a later layer) run the reference sequence followed by the separate kernels."""
        if self.fused:
            return self._run_fused(depth_image, labels_image, scale_factor, bool(flip_x), color_image)
        if getattr(self, "_hand_scratch", None) is None:   # once: nothing is allocated per call (or inside a graph capture)
            from .cuda.points_ops import PointsOps
            self._hand_scratch = (PointsOps(), GpuBuffer(self.labels_dims, dtype=np.uint16))
        po, tmp = self._hand_scratch
        ldims = np.array([self.labels_dims[1], self.labels_dims[0]], dtype=np.int32)
        if flip_x:
            self.run(depth_image, tmp, scale_factor)
            po.flip_x(ldims, tmp.cu(), labels_image.cu())
        else:
            self.run(depth_image, labels_image, scale_factor)
        if color_image is not None:
            po.make_rgba_from_labels(np.uint32(self.labels_dims[1]), np.uint32(self.labels_dims[0]),
                                     np.uint32(self.num_layered_classes), labels_image.cu(), self.label_colors.cu(),
                                     color_image.cu())

    def _run_fused(self, depth_image, labels_image, scale_factor, flip_x=False, color_image=None):
        """Same result through ONE C-ABI call (rdf_layered_run): the three fills are fused into the kernels."""
        import ctypes
        n = self.num_models
        if self._fused_args is None:
            vp, ci = ctypes.c_void_p * n, ctypes.c_int * n
            self._fused_args = dict(
                forests=vp(*[device_ptr(m.forest_cu) for m, _, _ in self.m]),
                n_trees=ci(*[int(m.num_trees) for m, _, _ in self.m]),
                max_depth=ci(*[int(m.max_depth) for m, _, _ in self.m]),
                n_classes=ci(*[int(m.num_classes) for m, _, _ in self.m]),
                filter_layer=ci(*[-1 if f is None else int(f) for _, f, _ in self.m]),
                filter_class=ci(*[-1 if c is None else int(c) for _, _, c in self.m]),
                layer_labels=vp(*[device_ptr(b) for b in self.label_images]), vp=vp)
        fa = self._fused_args
        packed = None
        if self.eval.use_packed:
            tabs = [m.packed(scale_factor) if m.max_depth <= 27 else None for m, _, _ in self.m]
            packed = fa["vp"](*[t.ptr if t is not None else None for t in tabs])
        ev = self.eval
        common = (device_ptr(depth_image), int(self.depth_dims[1]), int(self.depth_dims[0]), n,
                  packed, fa["forests"], fa["n_trees"], fa["max_depth"], fa["n_classes"],
                  fa["filter_layer"], fa["filter_class"], fa["layer_labels"],
                  device_ptr(self.labels_images_ptrs_cu), device_ptr(self.labels_conditions_cu),
                  int(self.labels_conditions_cu.shape[0]), device_ptr(labels_image),
                  ev._composite_bad.ptr, int(self.labels_reduce), float(scale_factor))
        if flip_x or color_image is not None:
            rc = ev._lib.rdf_layered_run_hand(*common, 1 if flip_x else 0, device_ptr(self.label_colors),
                                              int(self.num_layered_classes),
                                              device_ptr(color_image) if color_image is not None else None, ev._rt.stream())
            _lib.check(ev._lib, rc, "rdf_layered_run_hand")
            if color_image is not None:
                _touch(color_image.cu())
        else:
            rc = ev._lib.rdf_layered_run(*common, ev._rt.stream())
            _lib.check(ev._lib, rc, "rdf_layered_run")
        for b in self.label_images:
            _touch(b.cu())
        _touch(labels_image.cu())


class DecisionTreeEvaluator:
    def __init__(self, use_packed=True):
        self._rt = get_runtime()
        self._lib = self._rt.lib
        self.use_packed = use_packed
        # pixels whose composite walk was invalid (the reference device-asserts, tree_eval.cu:246-247)
        self._composite_bad = DeviceArray((1,), np.int32).fill(0)

    # -- single tree: evaluate_image_using_tree ------------------------------------------------
    def get_labels(self, tree, depth_images_in, labels_out):
        num_images, dim_y, dim_x = depth_images_in.shape
        assert tuple(labels_out.shape) == (num_images, dim_y, dim_x)
        lib, st = self._lib, self._rt.stream()
        for i0, n in _image_chunks(num_images, dim_y * dim_x):
            _lib.check(lib, lib.rdf_eval_tree(
                _at(depth_images_in, i0, dim_y * dim_x * 2), n, dim_x, dim_y,
                device_ptr(tree.tree_out_cu), int(tree.max_depth), int(tree.num_classes),
                _at(labels_out, i0, dim_y * dim_x * 2), st), "rdf_eval_tree")
        _touch(labels_out)

    # -- forest: evaluate_image_using_forest ---------------------------------------------------
    def get_labels_forest(self, forest, depth_images_in, labels_out, labels_reduce=1, filter_images=None,
                          filter_images_class=None, scale_factor=1.):
        self._forest_call(forest, depth_images_in, labels_out, labels_reduce, filter_images, filter_images_class,
                          scale_factor, False)

    def get_labels_forest_split(self, forest, depth_images_in, labels_out, helper_stream, helper_cus, labels_reduce=1,
                                scale_factor=1., queue_tag=0, fill_untouched=False):
        """get_labels_forest as TWO launches that share one tile queue (rdf_eval_forest_packed_split): a main launch on the
        current stream -- meant to be a CU-masked one (rdf_stream_create_with_reserved_cus) -- and a helper launch on
        `helper_stream` (a raw hipStream_t handle) that the caller has made wait for whatever occupies the `helper_cus` compute
        units the main stream leaves alone: a multi-GPU step's RCCL gather (distributed.ShardedForestEvaluator) -- and for
        whatever wrote the frames and pre-filled `labels_out` (`helper.wait_stream(current)`); readers wait for both streams.  Packed forests,
        one C-ABI call (< 2^31 pixels).  Returns the helper launch's workgroups (0: the launch was not split).  Not in the
        reference (single-GPU)."""
        import ctypes
        num_images, dim_y, dim_x = depth_images_in.shape
        assert tuple(labels_out.shape) == (num_images, dim_y // labels_reduce, dim_x // labels_reduce)
        assert num_images * dim_y * dim_x <= _PIX_LIMIT, "split launches take one C-ABI call: split the batch"
        packed = forest.packed(scale_factor)
        assert packed is not None and forest.max_depth <= 27
        n_helper = ctypes.c_int(0)
        rc = self._lib.rdf_eval_forest_packed_split(
            device_ptr(depth_images_in), int(num_images), int(dim_x), int(dim_y), packed.ptr, device_ptr(forest.forest_cu),
            int(forest.num_trees), int(forest.max_depth), int(forest.num_classes), None, -1, device_ptr(labels_out),
            int(labels_reduce), 1 if fill_untouched else 0, self._rt.stream(), ctypes.c_void_p(int(helper_stream)),
            int(helper_cus), int(queue_tag), ctypes.byref(n_helper))
        _lib.check(self._lib, rc, "rdf_eval_forest_packed_split")
        _touch(labels_out)
        return int(n_helper.value)

    def get_labels_forest_filled(self, forest, depth_images_in, labels_out, labels_reduce=1, filter_images=None,
                                 filter_images_class=None, scale_factor=1.):
        """get_labels_forest with the caller's `labels_out.fill(MAX_UINT16)` folded into the same pass: every label pixel the
        evaluation leaves alone (no depth, filtered out) is written 65535.  Not in the reference, whose callers fill first
        (decision_tree.py:237-240)."""
        self._forest_call(forest, depth_images_in, labels_out, labels_reduce, filter_images, filter_images_class,
                          scale_factor, True)

    def _forest_call(self, forest, depth_images_in, labels_out, labels_reduce, filter_images, filter_images_class,
                     scale_factor, fill_untouched):
        num_images, dim_y, dim_x = depth_images_in.shape

        assert tuple(labels_out.shape) == (num_images, dim_y // labels_reduce, dim_x // labels_reduce)

        if filter_images is not None:
            assert filter_images_class is not None
            assert tuple(filter_images.shape) == tuple(labels_out.shape)

        lib, st = self._lib, self._rt.stream()
        filter_class = int(filter_images_class) if filter_images is not None else -1
        lpix = (dim_y // labels_reduce) * (dim_x // labels_reduce)
        packed = None
        if self.use_packed and hasattr(forest, "packed") and forest.max_depth <= 27:
            packed = forest.packed(scale_factor)
            self._maybe_tune(forest, depth_images_in, labels_reduce, scale_factor)
        for i0, n in _image_chunks(num_images, dim_y * dim_x):
            d = _at(depth_images_in, i0, dim_y * dim_x * 2)
            o = _at(labels_out, i0, lpix * 2)
            f = _at(filter_images, i0, lpix * 2) if filter_images is not None else None
            if packed is not None:
                fn = lib.rdf_eval_forest_packed_filled if fill_untouched else lib.rdf_eval_forest_packed
                rc = fn(d, n, dim_x, dim_y, packed.ptr, device_ptr(forest.forest_cu),
                        int(forest.num_trees), int(forest.max_depth), int(forest.num_classes),
                        f, filter_class, o, int(labels_reduce), st)
                _lib.check(lib, rc, "rdf_eval_forest_packed")
            else:
                if fill_untouched:      # (the reference-layout entry point has no fused fill)
                    _lib.check(lib, lib.rdf_fill_u16(o, n * lpix, 65535, st), "rdf_fill_u16")
                rc = lib.rdf_eval_forest(d, n, dim_x, dim_y, device_ptr(forest.forest_cu),
                                         int(forest.num_trees), int(forest.max_depth), int(forest.num_classes),
                                         f, filter_class, o, int(labels_reduce), float(scale_factor), st)
                _lib.check(lib, rc, "rdf_eval_forest")
        _touch(labels_out)

    evaluate = get_labels_forest  # the name BASELINE.json's north_star uses; not in the reference

    # forests from this size on (hot records) may or may not be served better by the deep blocks: see DecisionForest.tune
    AUTO_TUNE_HOT_BYTES = 32 << 20
    AUTO_TUNE_PIXELS = 8 * 480 * 848
    AUTO_TUNE_SAMPLE_FRAMES = 32
    AUTO_TUNE_SAMPLE_PIXELS = 1 << 26       # the sample is one C-ABI call (< 2^31 pixels) and its scratch label map stays small

    def _maybe_tune(self, forest, depth_images_in, labels_reduce, scale_factor):
        """Safety net for callers that never tune: the first batch-sized evaluation of a big packed forest WHOSE TABLE CARRIES NO
        CHOICE makes it by measurement (DecisionForest.tune on up to 32 of the batch's own frames: a few dozen extra launches and
        a scratch label map, once per packed table), logs one line (logger "rdf_hip") and goes on; a failure there is logged and
        the evaluation proceeds with the heap-order records.  It never fires for a table that was tuned (`tune()` at load time:
        the documented flow), adopted with a choice inside (`adopt_packed`), or set by hand; `RDF_AUTO_TUNE=0` in the environment
        or `auto_tune = False` on the evaluator turns it off; a stream being captured into a hipGraph is never tuned on."""
        if not getattr(self, "auto_tune", True) or os.environ.get("RDF_AUTO_TUNE", "1") == "0":
            return
        if not hasattr(forest, "tune") or not hasattr(self._lib, "rdf_forest_tune"):
            return
        s = float(np.float32(scale_factor))
        done = forest.__dict__.setdefault("_tuned", {})
        if s in done:
            return
        n, h, w = (int(v) for v in depth_images_in.shape)
        if (int(forest.num_trees) << int(forest.max_depth)) * 16 < self.AUTO_TUNE_HOT_BYTES or n * h * w < self.AUTO_TUNE_PIXELS:
            return
        try:
            import torch
            if torch.cuda.is_current_stream_capturing():
                return
        except Exception:       # noqa: BLE001 -- (no torch runtime behind this evaluator: the host test double)
            return
        what = f"T{int(forest.num_trees)}/D{int(forest.max_depth)}/C{int(forest.num_classes)} forest"
        try:
            have = forest.deep_from(scale_factor) if hasattr(forest, "deep_from") else None
            if have is not None:        # the table carries a choice (tuned earlier, adopted, or set by hand): nothing to do
                done[s] = {"deep_from": have, "tried": None, "source": "the packed table"}
                return
            k = max(1, min(self.AUTO_TUNE_SAMPLE_FRAMES, n, self.AUTO_TUNE_SAMPLE_PIXELS // max(1, h * w)))
            t0 = time.perf_counter()
            done[s] = forest.tune(depth_images_in[0:k], labels_reduce, scale_factor)
            _log.warning("rdf: auto-tuned the deep-level table of a %s on %d of the batch's %dx%d frames in %.0f ms: deep_from=%d "
                         "(once per packed table; tune() at load time, adopt_packed() or RDF_AUTO_TUNE=0 avoid it)",
                         what, k, w, h, (time.perf_counter() - t0) * 1e3, done[s]["deep_from"])
        except Exception as e:      # noqa: BLE001 -- tuning is an optimisation: the evaluation itself must not fail because of it
            done[s] = {"deep_from": None, "tried": None, "error": f"{type(e).__name__}: {e}"[:300]}
            _log.warning("rdf: auto-tuning the deep-level table of a %s failed (%s); evaluating from the heap-order records",
                         what, done[s]["error"])

    # -- composite: make_composite_labels_image ------------------------------------------------
    def make_composite_labels_image(self, images, dim_x, dim_y, labels_decision_tree, composite_image):
        lib = self._lib
        n_cond = int(labels_decision_tree.shape[0])
        rc = lib.rdf_composite(device_ptr(images), int(images.shape[0]), int(dim_x), int(dim_y),
                               device_ptr(labels_decision_tree), n_cond, device_ptr(composite_image),
                               self._composite_bad.ptr, self._rt.stream())
        _lib.check(lib, rc, "rdf_composite")
        _touch(composite_image)

    def composite_bad_pixels(self, reset=True):
        """Pixels (since the last reset) whose composite walk left the conditions table.  Synchronises."""
        n = int(self._composite_bad.get()[0])
        if reset and n:
            self._composite_bad.fill(0)
        return n


# ---- training ---------------------------------------------------------------------------------
FEATURE_MAGNITUDE_MAX = 14.   # decision_tree.py:353
FEATURE_THRESHOLD_MAX = 11.   # _MIN = -_MAX


def make_random_offset():
    f_theta = np.random.uniform(0, np.pi * 2)
    magnitude = np.power(np.e, np.random.uniform(0, FEATURE_MAGNITUDE_MAX))   # linear in log space
    return np.array([np.cos(f_theta), np.sin(f_theta)]) * magnitude


def make_random_feature():
    return make_random_offset(), make_random_offset()


def make_random_threshold():
    return np.random.choice([-1, 1]) * np.power(np.e, np.random.uniform(0, FEATURE_THRESHOLD_MAX))


def make_random_features_loop(n, arr):
    """n proposals (ux, uy, vx, vy, thresh) drawn from the global numpy RNG exactly as the reference draws them
    (decision_tree.py:356-371), one Python call per draw, written into arr[n, 5] float32."""
    rows = []
    for _ in range(n):
        (u, v), t = make_random_feature(), make_random_threshold()
        rows.append((u[0], u[1], v[0], v[1], t))
    arr[:] = np.array(rows, dtype=np.float32)


def make_random_features(n, arr):
    """Same proposals as make_random_features_loop and the same state of the global RNG afterwards, without 6n
    Python-level draws (they were a quarter of the training time).  The global RandomState is MT19937: a
    uniform() consumes two 32-bit outputs (a = out >> 5, b = out >> 6, (a * 2^26 + b) / 2^53), choice([-1, 1])
    one (its low bit); a proposal is therefore 11 consecutive outputs, taken here in one randint call and put
    together with exact float64 arithmetic; cos, sin and power are the same numpy functions on arrays.
    tests/test_training.py checks values and RNG state against the loop."""
    if n == 0:
        return
    raw = np.random.randint(0, 1 << 32, size=11 * n, dtype=np.uint32).reshape(n, 11).astype(np.uint64)

    def dbl(k):   # the uniform [0, 1) double made of outputs k, k+1
        return ((raw[:, k] >> np.uint64(5)).astype(np.float64) * 67108864.0 +
                (raw[:, k + 1] >> np.uint64(6)).astype(np.float64)) / 9007199254740992.0

    def offset(k):
        theta = 0.0 + (np.pi * 2 - 0.0) * dbl(k)
        mag = np.power(np.e, 0.0 + (FEATURE_MAGNITUDE_MAX - 0.0) * dbl(k + 2))
        return np.cos(theta) * mag, np.sin(theta) * mag

    ux, uy = offset(0)
    vx, vy = offset(4)
    sign = np.array([-1, 1])[(raw[:, 8] & np.uint64(1)).astype(np.int64)]
    thr = sign * np.power(np.e, 0.0 + (FEATURE_THRESHOLD_MAX - 0.0) * dbl(9))
    arr[:] = np.stack([ux, uy, vx, vy, thr], axis=1).astype(np.float32)


class DecisionTreeTrainer:
    """Level-synchronous trainer of one tree; constructor, allocate() and train() as in the reference
    (decision_tree.py:373-600).  The training set stays resident in HBM (288 GB per GPU), so there are no
    image blocks to stream and no nvcomp-compressed node maps: NUM_IMAGES_PER_IMAGE_BLOCK is accepted and
    only checked for divisibility.  Proposal blocks and node blocks are kept: they decide which proposals
    compete and are therefore part of the result."""

    MAX_NEXT_NODES_TO_COUNT_PER_BLOCK = 2 ** 17   # decision_tree.py:424
    SORTED_ROWS_FROM_ACTIVE_NODES = 16            # levels with fewer active nodes count with the histogram kernel

    def __init__(self, NUM_IMAGES_PER_IMAGE_BLOCK, NUM_PROPOSALS_PER_PROPOSAL_BLOCK):
        self._rt = get_runtime()
        self._lib = self._rt.lib
        self.NUM_IMAGES_PER_IMAGE_BLOCK = NUM_IMAGES_PER_IMAGE_BLOCK
        self.NUM_PROPOSALS_PER_PROPOSAL_BLOCK = NUM_PROPOSALS_PER_PROPOSAL_BLOCK

    def allocate(self, dataset, NUM_RANDOM_FEATURES, MAX_TREE_DEPTH):
        self.NUM_RANDOM_FEATURES = NUM_RANDOM_FEATURES
        self.MAX_TREE_DEPTH = MAX_TREE_DEPTH
        per_block = self.NUM_IMAGES_PER_IMAGE_BLOCK or dataset.num_images
        assert dataset.num_images % per_block == 0
        assert self.NUM_RANDOM_FEATURES % self.NUM_PROPOSALS_PER_PROPOSAL_BLOCK == 0
        self.NUM_PROPOSAL_BLOCKS = self.NUM_RANDOM_FEATURES // self.NUM_PROPOSALS_PER_PROPOSAL_BLOCK
        C = dataset.num_classes()
        _, self.MAX_LEAF_NODES, _ = DecisionTree.get_config(MAX_TREE_DEPTH, C)
        P = self.NUM_PROPOSALS_PER_PROPOSAL_BLOCK
        shape = dataset.images_shape()
        assert shape[0] * shape[1] * shape[2] < _PIX_LIMIT, "training set too large for one call"

        self.node_counts_cu = DeviceArray((self.MAX_LEAF_NODES, C), np.uint64)
        self.next_node_counts_cu = DeviceArray((self.MAX_LEAF_NODES, C), np.uint64)
        self.active_nodes_cu = DeviceArray((self.MAX_LEAF_NODES,), np.int32)
        self.next_active_nodes_cu = DeviceArray((self.MAX_LEAF_NODES,), np.int32)
        self.next_num_active_nodes_cu = DeviceArray((1,), np.int32)
        self.get_next_num_active_nodes = lambda: int(self.next_num_active_nodes_cu.get()[0])
        self.best_gain_seen_per_node = DeviceArray((self.MAX_LEAF_NODES,), np.float32)
        self.current_proposals_block_cpu = np.zeros((P, 5), dtype=np.float32)
        self.current_proposals_block = DeviceArray((P, 5), np.float32)
        self.nodes_per_block = min(self.MAX_LEAF_NODES, self.MAX_NEXT_NODES_TO_COUNT_PER_BLOCK)
        self.current_next_node_counts_by_feature_cu_block = DeviceArray((P, self.nodes_per_block, C), np.uint64)
        ws = int(self._lib.rdf_train_histogram_workspace_bytes(P, self.nodes_per_block, C))
        self._hist_workspace = DeviceArray((max(ws, 8),), np.uint8).fill(0)   # the histogram calls keep it zero

        # the whole training set, resident
        self.depth_cu = DeviceArray(shape, np.uint16)
        self.labels_cu = DeviceArray(shape, np.uint16)
        self.nodes_by_pixel_cu = DeviceArray(shape, np.int32)
        ipb = dataset.images_per_block
        for b in range(dataset.num_images // ipb):
            dataset.get_depth_block_cu(b, self.depth_cu[b * ipb:(b + 1) * ipb])
            dataset.get_labels_block_cu(b, self.labels_cu[b * ipb:(b + 1) * ipb])

        # Counting by sorted rows (rdf_train_sort_pixels / _decision_bits / _count_rows): one row of P bits per labelled
        # pixel, rows in (node, class) order.  Used for proposal blocks of up to 1024 proposals; bigger blocks keep the
        # histogram kernel with its per-wave atomics.  `use_sorted_rows = False` forces the latter (tests compare the two).
        row_bytes = int(self._lib.rdf_train_bits_row_bytes(P))
        self.use_sorted_rows = row_bytes > 0
        if self.use_sorted_rows:
            t = self.labels_cu.torch_bytes().view(self._rt.torch.int16)
            n_labelled = int((t != 0).sum().item())
            self.pixel_rows_cu = DeviceArray(shape, np.int32)
            self.row_keys_cu = DeviceArray((max(n_labelled, 1),), np.int32)
            self.decision_bits_cu = DeviceArray((max(n_labelled, 1) * row_bytes,), np.uint8)
            self._sort_workspace = DeviceArray((int(self._lib.rdf_train_sort_workspace_bytes(self.MAX_LEAF_NODES, C)),), np.uint8)
            self._bits_workspace = DeviceArray((int(self._lib.rdf_train_bits_workspace_bytes(P)),), np.uint8)

    def train(self, dataset, tree):
        lib, st = self._lib, self._rt.stream
        C = dataset.num_classes()
        n_img, dim_y, dim_x = self.depth_cu.shape
        D = self.MAX_TREE_DEPTH
        P = self.NUM_PROPOSALS_PER_PROPOSAL_BLOCK
        NB = self.nodes_per_block
        chk = lambda rc, what: _lib.check(lib, rc, what)

        tree.tree_out_cu.fill(np.float32(0.))
        self.node_counts_cu.fill(0)
        self.next_node_counts_cu.fill(0)
        chk(lib.rdf_train_init(self.labels_cu.ptr, self.labels_cu.size, C, self.nodes_by_pixel_cu.ptr,
                               self.node_counts_cu.ptr, st()), "rdf_train_init")
        self.next_node_counts_cu.copy_from(self.node_counts_cu)   # both start from the root counts (:399-400)

        self.active_nodes_cu.fill(np.int32(0))          # one node to start, index 0
        self.next_num_active_nodes_cu.fill(np.int32(1))

        self.level_seconds = []      # (filled when self.time_levels is set: one synchronisation per level)
        time_levels = bool(getattr(self, "time_levels", False))
        for current_level in range(D):
            num_active_nodes = self.get_next_num_active_nodes()
            if num_active_nodes == 0:
                break
            if time_levels:
                import time
                self._rt.synchronize()
                t_level = time.perf_counter()
            self.best_gain_seen_per_node.fill(np.float32(-1.))
            # (levels with a handful of nodes keep the histogram kernel: a wave meets one or two groups there, and the
            # sort is work it does not have -- 30 against 32 ms per level on the 256-frame benchmark, 45 against 36 at 64 nodes)
            sorted_rows = self.use_sorted_rows and num_active_nodes >= self.SORTED_ROWS_FROM_ACTIVE_NODES
            n_level_nodes = 2 ** current_level
            if sorted_rows:      # once per level: every live pixel's row in (node, class) order
                chk(lib.rdf_train_sort_pixels(self.labels_cu.ptr, self.nodes_by_pixel_cu.ptr, self.labels_cu.size, C,
                                              n_level_nodes, self.pixel_rows_cu.ptr, self.row_keys_cu.ptr,
                                              self._sort_workspace.ptr, st()), "rdf_train_sort_pixels")

            for _ in range(self.NUM_PROPOSAL_BLOCKS):
                make_random_features(P, self.current_proposals_block_cpu)
                self.current_proposals_block.set(self.current_proposals_block_cpu)
                if sorted_rows:  # once per proposal block: the decisions of every live pixel (they do not depend on the node block)
                    chk(lib.rdf_train_decision_bits(self.depth_cu.ptr, self.pixel_rows_cu.ptr, n_img, dim_x, dim_y,
                                                    self.current_proposals_block.ptr, P, self.decision_bits_cu.ptr,
                                                    self._bits_workspace.ptr, st()),
                        "rdf_train_decision_bits")

                max_active_nodes_next_level = 2 ** (current_level + 1)
                if max_active_nodes_next_level > self.MAX_NEXT_NODES_TO_COUNT_PER_BLOCK:
                    n_node_blocks = max_active_nodes_next_level // self.MAX_NEXT_NODES_TO_COUNT_PER_BLOCK
                    node_blocks = [(i * self.MAX_NEXT_NODES_TO_COUNT_PER_BLOCK, (i + 1) * self.MAX_NEXT_NODES_TO_COUNT_PER_BLOCK)
                                   for i in range(n_node_blocks)]
                else:
                    node_blocks = [(0, max_active_nodes_next_level)]

                for node_block_start, node_block_end in node_blocks:
                    self.current_next_node_counts_by_feature_cu_block.fill(0)
                    # evaluate_random_features, as two calls: the kernel counts the left children, the right ones
                    # follow from the parents' counts (the whole training set is counted in this one call)
                    if sorted_rows:
                        chk(lib.rdf_train_count_rows(self.decision_bits_cu.ptr, self.row_keys_cu.ptr, self._sort_workspace.ptr,
                                                     n_level_nodes, P, C, node_block_start, node_block_end, NB,
                                                     self.current_next_node_counts_by_feature_cu_block.ptr, st()),
                            "rdf_train_count_rows")
                    else:
                        chk(lib.rdf_train_histogram_left_ws(self.depth_cu.ptr, self.labels_cu.ptr, self.nodes_by_pixel_cu.ptr,
                                                            n_img, dim_x, dim_y, self.current_proposals_block.ptr, P, C,
                                                            node_block_start, node_block_end, NB,
                                                            self.current_next_node_counts_by_feature_cu_block.ptr,
                                                            self._hist_workspace.ptr, self.node_counts_cu.ptr, st()),
                            "rdf_train_histogram_left_ws")
                    chk(lib.rdf_train_right_counts(num_active_nodes, self.active_nodes_cu.ptr, P, NB, node_block_start,
                                                   node_block_end, C, self.node_counts_cu.ptr,
                                                   self.current_next_node_counts_by_feature_cu_block.ptr, st()),
                        "rdf_train_right_counts")
                    chk(lib.rdf_train_pick_best(num_active_nodes, self.active_nodes_cu.ptr, P, D, NB, node_block_start,
                                                node_block_end, C, current_level, self.node_counts_cu.ptr,
                                                self.current_next_node_counts_by_feature_cu_block.ptr,
                                                self.current_proposals_block.ptr, device_ptr(tree.tree_out_cu),
                                                self.next_node_counts_cu.ptr, self.best_gain_seen_per_node.ptr, st()),
                        "rdf_train_pick_best")

            self.next_num_active_nodes_cu.fill(np.int32(0))
            self.next_active_nodes_cu.fill(np.int32(0))
            chk(lib.rdf_train_next_active(current_level, D, C, device_ptr(tree.tree_out_cu), self.active_nodes_cu.ptr,
                                          num_active_nodes, self.next_active_nodes_cu.ptr,
                                          self.next_num_active_nodes_cu.ptr, st()), "rdf_train_next_active")
            if time_levels:
                self._rt.synchronize()
                self.level_seconds.append((current_level, num_active_nodes, time.perf_counter() - t_level))
            if current_level == D - 1:
                break
            self.node_counts_cu.copy_from(self.next_node_counts_cu)
            chk(lib.rdf_train_update_pixels(self.depth_cu.ptr, n_img, dim_x, dim_y, current_level, D, C,
                                            self.nodes_by_pixel_cu.ptr, device_ptr(tree.tree_out_cu), st()),
                "rdf_train_update_pixels")
            self.active_nodes_cu.copy_from(self.next_active_nodes_cu)
        _touch(tree.tree_out_cu)


def _image_chunks(num_images, pix_per_image):
    """Split a batch so that one C-ABI call addresses < 2^31 depth pixels."""
    if num_images <= 0 or pix_per_image <= 0:
        return [(0, max(int(num_images), 0))]
    per = max(1, _PIX_LIMIT // int(pix_per_image))
    return [(i, min(per, num_images - i)) for i in range(0, num_images, per)]


def _at(arr, first_image, bytes_per_image):
    return device_ptr(arr) + int(first_image) * int(bytes_per_image)


def _touch(arr):
    if isinstance(arr, DeviceArray):
        arr.mark_dirty()
