"""Parity of the HIP path against the CPU oracle -- runs on the MI355X box (`-m gpu`).

Everything goes through the package's reference-compatible API, i.e. through the C ABI of
librdf_hip.so.  Bit-exact is the bar: label maps are integers."""
import json
import os

import numpy as np
import pytest

import kat_cases

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "rdf_golden_v1.npz")
CASES = kat_cases.cases()
COMPOSITE = kat_cases.composite_cases()


class _Tree:  # duck type of DecisionTree for arbitrary device trees
    def __init__(self, rdf, t):
        n, e = t.shape
        self.max_depth = int(np.log2(n + 1))
        self.num_classes = (e - 7) // 2
        self.tree_out_cu = rdf.to_device(t)


def _gpu_forest(rdf, ev, depth, forest, prefill, r=1, filt=None, fcls=None, s=1.0):
    n, h, w = depth.shape
    f = rdf.DecisionForest.from_numpy(forest)
    out = rdf.DeviceArray((n, h // r, w // r), np.uint16).fill(prefill)
    ev.get_labels_forest(f, rdf.to_device(depth), out, r, rdf.to_device(filt) if filt is not None else None,
                         fcls, s)
    return out.get()


@pytest.fixture(scope="module")
def evs(rdf, gpu_runtime):
    return {"packed": rdf.DecisionTreeEvaluator(use_packed=True), "direct": rdf.DecisionTreeEvaluator(use_packed=False)}


def test_native_library_is_the_one_loaded(rdf, gpu_runtime):
    """The HIP library is mapped into this process AND was built from the sources next to it: its baked-in build id
    (rdf_build_id: a hash of the .hip files, the headers and the compiler flags) equals the one recomputed here."""
    from importlib import import_module
    assert gpu_runtime.name == "hip"
    maps = open("/proc/self/maps").read()
    assert "librdf_hip.so" in maps
    build = import_module("3d-beats_amd._build")
    got = gpu_runtime.lib.rdf_build_id()
    got = got.decode() if isinstance(got, bytes) else got
    assert len(got) == 16 and got == build.source_id() == build.built_id(), (got, build.source_id(), build.built_id())


def test_float_helpers_match_ieee(rdf, gpu_runtime):
    lib, st = gpu_runtime.lib, gpu_runtime.stream()
    x = np.array([0.0, -0.0, 0.5, -0.5, -1.0, -1.5, 1.999, 2147483520.0, 2147483648.0, 3e9, -2147483648.0,
                  -2147483904.0, -3e9, np.inf, -np.inf, np.nan, 1e-45, -1e-45, 65535.7, -65535.2], dtype=np.float32)
    want = np.array([0, 0, 0, -1, -1, -2, 1, 2147483520, 2147483647, 2147483647, -2147483648, -2147483648,
                     -2147483648, 2147483647, -2147483648, 0, 0, -1, 65535, -65536], dtype=np.int64).astype(np.int32)
    dx, do = rdf.to_device(x), rdf.DeviceArray(x.shape, np.int32)
    assert lib.rdf_debug_floor_i32(dx.ptr, do.ptr, x.size, st) == 0
    assert np.array_equal(do.get(), want)
    rng = np.random.default_rng(1)
    num = np.concatenate([(rng.standard_normal(200000) * np.exp(rng.uniform(0, 14, 200000))),
                          [0.0, -0.0, 1e-40, -1e-40, 3e38, -3e38, np.inf, -np.inf, 1e-45, 1.17549435e-38]]).astype(np.float32)
    den = rng.integers(1, 65535, size=num.size).astype(np.float32)
    dn, dd, dq = rdf.to_device(num), rdf.to_device(den), rdf.DeviceArray(num.shape, np.float32)
    assert lib.rdf_debug_div_f32(dn.ptr, dd.ptr, dq.ptr, num.size, st) == 0
    with np.errstate(all="ignore"):
        ref = num / den
    assert np.array_equal(dq.get().view(np.uint32), ref.view(np.uint32))


@pytest.mark.parametrize("path", ["packed", "direct"])
@pytest.mark.parametrize("prefill", [65535, 0])
@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_known_answers_on_gpu(case, prefill, path, rdf, evs):
    ev = evs[path]
    if case["kind"] == "forest":
        got = _gpu_forest(rdf, ev, case["depth"], case["forest"], prefill, case["labels_reduce"], case["filter"],
                          case["filter_class"], case["scale_factor"])
    else:
        out = rdf.DeviceArray(case["depth"].shape, np.uint16).fill(prefill)
        ev.get_labels(_Tree(rdf, case["tree"]), rdf.to_device(case["depth"]), out)
        got = out.get()
    want = kat_cases.expected_array(case["expected"], prefill)
    assert np.array_equal(got, want), f"{case['name']} ({case['cite']}):\n got {got}\nwant {want}"


@pytest.mark.parametrize("case", COMPOSITE, ids=[c["name"] for c in COMPOSITE])
def test_composite_known_answers_on_gpu(case, rdf, evs):
    ev = evs["packed"]
    for prefill in (65535, 0):
        imgs = [rdf.to_device(a) for a in case["images"]]
        table = rdf.to_device(np.array([i.ptr for i in imgs], dtype=np.int64))
        h, w = case["images"][0].shape
        out = rdf.DeviceArray((1, h, w), np.uint16).fill(prefill)
        ev.make_composite_labels_image(table, w, h, rdf.to_device(case["cond"]), out)
        assert np.array_equal(out.get(), kat_cases.expected_array(case["expected"], prefill)), case["name"]
        assert ev.composite_bad_pixels() == case["bad"]


@pytest.mark.parametrize("path", ["packed", "direct"])
def test_golden_vectors_on_gpu(path, rdf, evs):
    g = np.load(GOLDEN)
    ev = evs[path]
    assert np.array_equal(_gpu_forest(rdf, ev, g["g1_depth"], g["g1_forest"], 65535), g["g1_labels"])
    assert np.array_equal(_gpu_forest(rdf, ev, g["g2_depth"], g["g2_forest"], 0, 2, g["g2_filter"], 1, 0.5),
                          g["g2_labels"])
    out = rdf.DeviceArray(g["g4_labels"].shape, np.uint16).fill(7)
    ev.get_labels(_Tree(rdf, g["g4_tree"]), rdf.to_device(g["g4_depth"]), out)
    assert np.array_equal(out.get(), g["g4_labels"])


def test_layered_run_on_gpu_matches_golden(rdf, gpu_runtime, tmp_path):
    g = np.load(GOLDEN)
    np.save(tmp_path / "l0.npy", g["g3_forest0"])
    np.save(tmp_path / "l1.npy", g["g3_forest1"])
    cfg = {"layers": [{"model": "l0.npy"}, {"model": "l1.npy", "filter_model": 0, "filter_model_class": 3}],
           "conditions": g["g3_cond"].tolist(), "label_colors": [[1, 2, 3, 4]] * 4}
    (tmp_path / "cfg.json").write_text(json.dumps(cfg))
    lf = rdf.LayeredDecisionForest.load(str(tmp_path / "cfg.json"), (60, 84), labels_reduce=2)
    depth, labels = rdf.GpuBuffer((60, 84), np.uint16), rdf.GpuBuffer((30, 42), np.uint16)
    depth.cu().set(g["g3_depth"][0])
    for fused in (True, False, True):  # one-call path, reference-like sequence, and again (buffers re-filled)
        lf.fused = fused
        labels.cu().fill(12345)
        lf.run(depth, labels, 1.0)
        assert np.array_equal(lf.label_images[0].cu().get(), g["g3_l0"][0])
        assert np.array_equal(lf.label_images[1].cu().get(), g["g3_l1"][0])
        assert np.array_equal(labels.cu().get(), g["g3_comp"][0])
    assert lf.eval.composite_bad_pixels() == 0


@pytest.mark.parametrize("kind", ["live", "dense"])
def test_config3_full_size_plain_layered_run(kind, rdf, gpu_runtime, oracle):
    """BASELINE configs[2] at its full size through the plain drop-in call: LayeredDecisionForest.run on one
    848x480 frame, labels_reduce 2, scale 848/848, a 2-layer stack whose second layer filters on the first one's class 3
    (run_live_layered.py:54, 126) -- the fused one-call route (layers in one launch, a wave per tree), the same with those
    two switched off, and the reference's step-by-step sequence: every per-layer label image and the composite are the
    oracle's.  (tests/test_pipeline.py reaches this size only through HandPipeline.)"""
    lib = gpu_runtime.lib
    synth = rdf.synth
    f0, f1 = synth.forest(3, 12, 4, "trained", 80), synth.forest(4, 14, 4, "trained", 90)
    cond = [[0, 1], [0, 2], [1, 3], [0, 3], [0, 4], [0, 5]]          # layer 0: 1, 2 final, 3 -> layer 1 (decision_tree.py:214-220)
    cfg = {"layers": [{"model": rdf.DecisionForest.from_numpy(f0)},
                      {"model": rdf.DecisionForest.from_numpy(f1), "filter_model": 0, "filter_model_class": 3}],
           "conditions": cond, "label_colors": [[9, 8, 7, 255]] * 5}
    H, W, R = 480, 848, 2
    frame = synth.frames([kind], 4100, H, W)
    l0 = np.full((1, H // R, W // R), 65535, np.uint16)
    l1, comp = l0.copy(), l0.copy()
    oracle.eval_forest(frame, f0, l0, R, None, None, 1.0)
    oracle.eval_forest(frame, f1, l1, R, l0, 3, 1.0)
    oracle.composite([l0[0], l1[0]], np.array(cond, np.int32), comp)
    assert (l1 != 65535).any() and len(np.unique(comp)) >= 4
    lf = rdf.LayeredDecisionForest(cfg, (H, W), R)
    depth, labels = rdf.GpuBuffer((H, W), np.uint16), rdf.GpuBuffer((H // R, W // R), np.uint16)
    depth.cu().set(frame[0])
    try:
        for fused, one_launch, tw in ((True, -1, -1), (True, 0, 0), (False, -1, -1), (True, -1, 0)):
            lf.fused = fused
            lib.rdf_set_layers_one_launch(one_launch)
            lib.rdf_set_tree_waves(tw)
            labels.cu().fill(4242)
            lf.run(depth, labels, 848 / 848)
            what = (fused, one_launch, tw)
            assert np.array_equal(lf.label_images[0].cu().get(), l0[0]), what
            assert np.array_equal(lf.label_images[1].cu().get(), l1[0]), what
            assert np.array_equal(labels.cu().get(), comp[0]), what
        assert lf.eval.composite_bad_pixels() == 0
    finally:
        lib.rdf_set_layers_one_launch(-1)
        lib.rdf_set_tree_waves(-1)


SWEEP = [
    # T, D, C, topo, n, h, w, r, s, filter
    (1, 5, 2, "trained", 1, 17, 23, 1, 1.0, False),
    (2, 12, 3, "trained", 2, 61, 67, 1, 1.0, True),
    (3, 11, 4, "full", 3, 40, 130, 2, 0.5, False),
    (4, 13, 4, "trained", 2, 96, 160, 1, 1.0, False),
    (5, 10, 5, "trained", 2, 50, 77, 3, 2.0, True),
    (7, 9, 7, "full", 1, 64, 64, 1, 0.75, False),
    (8, 12, 9, "trained", 2, 48, 80, 2, 1.0, False),
    (9, 7, 17, "trained", 1, 33, 65, 1, 1.0, False),
    (4, 10, 33, "full", 1, 20, 70, 1, 1.0, True),
    (4, 1, 4, "full", 1, 9, 11, 1, 1.0, False),
    (6, 10, 4, "trained", 2, 70, 131, 1, 1.0, False),   # two passes of the 3-wide kernel
    (1, 14, 4, "full", 2, 64, 200, 1, 1.0, True),       # 1-wide kernel, deep, filtered
    (2, 13, 5, "trained", 1, 90, 90, 2, 1.5, False),    # 2-wide kernel, 8-class accumulator
]


@pytest.mark.parametrize("path", ["packed", "direct"])
@pytest.mark.parametrize("cfg", SWEEP, ids=[f"T{c[0]}D{c[1]}C{c[2]}{c[3]}r{c[7]}" for c in SWEEP])
def test_random_sweep_bit_exact(cfg, path, rdf, evs, oracle):
    T, D, C, topo, n, h, w, r, s, use_filter = cfg
    synth = rdf.synth
    forest = synth.forest(T, D, C, topo, first_tree=T * 3)
    depth = synth.frames((["dense", "live"] * n)[:n], 500 + T, h, w)
    filt = (np.random.default_rng(T).integers(0, 3, size=(n, h // r, w // r))).astype(np.uint16) if use_filter else None
    want = np.full((n, h // r, w // r), 65535, np.uint16)
    oracle.eval_forest(depth, forest, want, r, filt, 1 if use_filter else None, s)
    got = _gpu_forest(rdf, evs[path], depth, forest, 65535, r, filt, 1 if use_filter else None, s)
    assert np.array_equal(got, want), f"{(got != want).sum()} of {want.size} pixels differ"


@pytest.mark.parametrize("block,lds,halo,sched", [(256, 0, 16, 1), (256, 4096, 0, 1), (512, 81920, 16, 0),
                                                  (1024, 163840, 40, 1), (1024, 0, 3, 0), (256, 81920, 200, 1),
                                                  (512, 40000, 7, 1), (256, 0, 24, 2), (512, 20000, 8, 2)])
def test_launch_geometry_does_not_change_results(block, lds, halo, sched, rdf, gpu_runtime, evs, oracle):
    """Workgroup size, LDS budget (levels held in LDS), staged-tile halo (incl. 'tile does not fit')
    and the tile schedule (static, dynamic queue, one tile per workgroup) are performance knobs only."""
    synth = rdf.synth
    forest = synth.forest(4, 12, 4, "trained")
    depth = synth.frames(["live", "dense", "live"], 700, 120, 200)
    lib = gpu_runtime.lib
    lib.rdf_set_block_threads(block)
    lib.rdf_set_lds_budget_bytes(lds if lds else 1)
    lib.rdf_set_halo(halo)
    lib.rdf_set_scheduler(sched)
    try:
        for r, s in ((1, 1.0), (2, 0.5), (5, 1.0)):
            want = np.full((3, 120 // r, 200 // r), 65535, np.uint16)
            oracle.eval_forest(depth, forest, want, r, scale_factor=s)
            for path in ("packed", "direct"):
                got = _gpu_forest(rdf, evs[path], depth, forest, 65535, r, s=s)
                assert np.array_equal(got, want), (block, lds, halo, sched, path, r)
    finally:
        lib.rdf_set_block_threads(0)
        lib.rdf_set_lds_budget_bytes(0)
        lib.rdf_set_halo(-1)
        lib.rdf_set_scheduler(-1)


@pytest.mark.parametrize("block,lds,halo,levels,vec", [(256, 0, 24, -1, 1), (256, 0, 24, -1, 0), (256, 0, 16, 3, 1),
                                                      (1024, 163840, 96, 8, 1), (1024, 163840, 88, 6, 1),
                                                      (512, 81920, 64, 8, 1), (512, 81920, 40, 0, 1),
                                                      (1024, 163840, 200, 8, 1), (256, 20000, 8, 12, 1)])
def test_vector_staging_and_pinned_levels_do_not_change_results(block, lds, halo, levels, vec, rdf, gpu_runtime, evs, oracle):
    """16-byte tile staging (frame width and halo multiples of 8), also with the forest's top levels pinned in LDS and the
    depth tile taking the rest of the budget (big halos, one workgroup per CU), equals the oracle; so does a frame whose
    width rules the vector path out."""
    synth = rdf.synth
    forest = synth.forest(4, 12, 4, "trained", first_tree=11)
    lib = gpu_runtime.lib
    lib.rdf_set_block_threads(block)
    lib.rdf_set_lds_budget_bytes(lds if lds else 1)
    lib.rdf_set_halo(halo)
    lib.rdf_set_lds_levels(levels)
    lib.rdf_set_stage_vec(vec)
    try:
        for (h, w) in ((136, 200), (97, 72), (130, 203)):
            depth = synth.frames(["live", "dense", "live"], 710, h, w)
            depth[1, :3, :] = 65535
            depth[1, :, -9:] = 0
            for r, s in ((1, 1.0), (2, 0.5), (4, 2.0)):
                filt = (np.random.default_rng(h + r).integers(0, 3, size=(3, h // r, w // r))).astype(np.uint16)
                for use_filter in (False, True):
                    want = np.full((3, h // r, w // r), 65535, np.uint16)
                    oracle.eval_forest(depth, forest, want, r, filt if use_filter else None, 1 if use_filter else None, s)
                    for path in ("packed", "direct"):
                        got = _gpu_forest(rdf, evs[path], depth, forest, 65535, r, filt if use_filter else None,
                                          1 if use_filter else None, s)
                        assert np.array_equal(got, want), (block, lds, halo, levels, vec, h, w, path, r, use_filter)
    finally:
        lib.rdf_set_block_threads(0)
        lib.rdf_set_lds_budget_bytes(0)
        lib.rdf_set_halo(-1)
        lib.rdf_set_lds_levels(-1)
        lib.rdf_set_stage_vec(-1)


ADVERSARIAL = [8388607.5, 8388607.0, 8388608.0, -8388608.0, -8388608.5, -8388607.5, 16777216.0, 3e9, -3e9, 1e-30,
               -1e-30, 1e-45, -1e-45, 2.0 ** -87, -(2.0 ** -87), 2.0 ** -88, 11.999999, -11.999999, 23.999998,
               0.99999994, -0.99999994, -1.0000001, 65534.0, 65533.996, 131067.99, 4000.0, 3999.9998, -4000.0,
               0.0, -0.0, 0.5, -0.5, 1e38, -1e38, float("inf"), float("-inf"), float("nan"), 2.5e6, -2.5e6,
               7999.9995, 12345.678, -12345.678, 1.17549435e-38, 8388606.5,
               # edges of the 23-bit numerator field of the hot record (|a| < 2^22 stays on the fast path)
               4194303.5, 4194303.75, 4194304.0, -4194304.0, -4194303.5, -4194304.5, 4194302.0, -4194303.0, 2097151.9]


@pytest.mark.parametrize("path", ["packed", "direct"])
@pytest.mark.parametrize("force_exact", [0, 1])
def test_adversarial_numerators_and_exact_fallback(path, force_exact, rdf, gpu_runtime, oracle):
    """Offsets at the edges of what the 16-byte integer record can hold (|a| >= 2^22, denormals, inf, NaN,
    values a hair below integers) must take the IEEE branch and still match; force_exact=1 sends EVERY
    node of the packed path through that branch."""
    rng = np.random.default_rng(77)
    T, D, C = 4, 7, 4
    n = (1 << D) - 1
    forest = rdf.synth.forest(T, D, C, "trained", first_tree=50)
    adv = np.array(ADVERSARIAL, dtype=np.float32)
    pick = rng.random((T, n, 4)) < 0.5
    forest[:, :, 0:4] = np.where(pick, adv[rng.integers(0, adv.size, size=(T, n, 4))], forest[:, :, 0:4])
    forest[:, :, 4] = np.where(rng.random((T, n)) < 0.3,
                               np.array([np.nan, np.inf, -np.inf, 65535.5, -65535.0, -65534.5, 0.5, -0.5, 7.0, 1e9],
                                        dtype=np.float32)[rng.integers(0, 10, size=(T, n))], forest[:, :, 4])
    depth = rdf.synth.frames(["dense", "live"], 40, 70, 90)
    depth[0, :8, :8] = np.array([1, 2, 3, 7, 4000, 65534, 11, 12], dtype=np.uint16)[None, :]
    lib = gpu_runtime.lib
    lib.rdf_set_force_exact(force_exact)
    try:
        ev = rdf.DecisionTreeEvaluator(use_packed=(path == "packed"))
        for r, s in ((1, 1.0), (2, 0.5), (1, 3.0)):
            want = np.full((2, 70 // r, 90 // r), 65535, np.uint16)
            oracle.eval_forest(depth, forest, want, r, scale_factor=s)
            got = _gpu_forest(rdf, ev, depth, forest, 65535, r, s=s)
            assert np.array_equal(got, want), f"{(got != want).sum()} pixels differ (r={r}, s={s})"
    finally:
        lib.rdf_set_force_exact(0)


@pytest.mark.parametrize("topology", ["full", "trained"])
def test_config2_full_size_bit_exact(topology, rdf, evs, oracle, gpu_runtime):
    """BASELINE config 2: 848x480 frames (one dense, one live-like), 4-tree depth-20 forest, C=4."""
    synth = rdf.synth
    forest = synth.forest(4, 20, 4, topology)
    depth = synth.frames(["dense", "live"], 0)
    want = np.full(depth.shape, 65535, np.uint16)
    st = np.zeros(3, np.uint64)
    oracle.eval_forest(depth, forest, want, stats=st)
    for path in ("packed", "direct"):
        got = _gpu_forest(rdf, evs[path], depth, forest, 65535)
        assert np.array_equal(got, want), f"{path}: {(got != want).sum()} pixels differ"
    # visit counters of the stats kernel == oracle's (they define the algorithmic bytes)
    lib = gpu_runtime.lib
    f = rdf.DecisionForest.from_numpy(forest)
    out = rdf.DeviceArray(depth.shape, np.uint16).fill(65535)
    dstats = rdf.DeviceArray((3,), np.uint64).fill(0)
    rc = lib.rdf_eval_forest_stats(rdf.to_device(depth).ptr, 2, 848, 480, f.forest_cu.ptr, 4, 20, 4, None, -1, out.ptr,
                                   1, 1.0, dstats.ptr, gpu_runtime.stream())
    assert rc == 0
    assert np.array_equal(out.get(), want)
    assert np.array_equal(dstats.get(), st)
    if topology == "full":
        assert oracle.order_sensitive(depth, forest) == 0


def test_batch_properties_at_scale(rdf, evs, gpu_runtime):
    """Size-independent properties on a 32-frame 848x480 batch (too big to run through the oracle in
    seconds): batching == frame-by-frame, idempotence, untouched pixels keep any pre-fill, packed == direct."""
    synth = rdf.synth
    forest = rdf.DecisionForest.from_numpy(synth.forest(4, 16, 4, "trained"))
    host = synth.mixed_batch(32, first_idx=900)
    depth = rdf.to_device(host)
    ev = evs["packed"]
    a = rdf.DeviceArray(host.shape, np.uint16).fill(65535)
    ev.get_labels_forest(forest, depth, a)
    A = a.get()
    ev.get_labels_forest(forest, depth, a)           # idempotent
    assert np.array_equal(a.get(), A)
    b = rdf.DeviceArray(host.shape, np.uint16).fill(1234)
    evs["direct"].get_labels_forest(forest, depth, b)
    B = b.get()
    invalid = (host == 0) | (host == 65535)
    assert (A[invalid] == 65535).all() and (B[invalid] == 1234).all()
    assert np.array_equal(A[~invalid], B[~invalid])
    assert A[~invalid].max() < 4
    for i in (0, 15, 16, 31):                        # frame-by-frame
        one = rdf.DeviceArray((1,) + host.shape[1:], np.uint16).fill(65535)
        ev.get_labels_forest(forest, depth[i:i + 1], one)
        assert np.array_equal(one.get()[0], A[i])


def test_config4_shard_bit_exact(rdf, evs, oracle, gpu_runtime):
    """BASELINE configs[3]'s per-GPU shard exactly as the benchmark runs it: 128 frames of 848x480, half
    dense and half live-like, T4/D20/C4 "full" forest, one launch -- EVERY frame against the oracle (a few seconds of
    oracle/rdf_oracle.c on the box's cores), with the visit counters, through the packed tables (heap-order records and deep
    blocks) and the reference layout.  bench.py makes the same comparison inside every default run; this puts it into -m gpu."""
    synth = rdf.synth
    forest_np = synth.forest(4, 20, 4, "full")
    host = synth.mixed_batch(128)
    assert host.shape == (128, 480, 848)
    want = np.full(host.shape, 65535, np.uint16)
    st = np.zeros(3, np.uint64)
    oracle.eval_forest(host, forest_np, want, stats=st)
    assert 0.4 * host.size < int(st[0]) < 0.7 * host.size and int(st[1]) == 20 * 4 * int(st[0])      # every walk reaches level D-1
    forest = rdf.DecisionForest.from_numpy(forest_np)
    depth = rdf.to_device(host)
    out = rdf.DeviceArray(host.shape, np.uint16)
    lib = gpu_runtime.lib
    try:
        for name, ev, deep in (("packed", evs["packed"], 0), ("packed, deep blocks from level 12", evs["packed"], 12), ("reference layout", evs["direct"], -1)):
            lib.rdf_set_deep_from(deep)
            out.fill(65535)
            ev.get_labels_forest(forest, depth, out)
            got = out.get()
            assert np.array_equal(got, want), (name, int((got != want).sum()))
        lib.rdf_set_deep_from(-1)
        # the launch's own counters (the roofline's algorithmic bytes come from them) equal the oracle's
        dstats = rdf.DeviceArray((8,), np.uint64).fill(0)
        out.fill(65535)
        rc = lib.rdf_eval_forest_packed_stats(depth.ptr, 128, 848, 480, forest.packed(1.0).ptr, forest.forest_cu.ptr, 4, 20, 4, out.ptr, 1,
                                              dstats.ptr, gpu_runtime.stream())
        assert rc == 0 and [int(v) for v in dstats.get()[0:3]] == [int(v) for v in st] and np.array_equal(out.get(), want)
    finally:
        lib.rdf_set_deep_from(-1)


def test_fill_u16_any_alignment(rdf, gpu_runtime):
    lib, st = gpu_runtime.lib, gpu_runtime.stream()
    buf = rdf.DeviceArray((5000,), np.uint16)
    for off, n in [(0, 5000), (1, 4999), (3, 17), (7, 1), (5, 0), (2, 4096), (9, 1001)]:
        buf.fill(0)
        assert lib.rdf_fill_u16(buf.ptr + 2 * off, n, 65535, st) == 0
        h = buf.get()
        assert (h[off:off + n] == 65535).all() and h[:off].sum() == 0 and h[off + n:].sum() == 0


def test_bad_arguments_are_rejected(rdf, gpu_runtime):
    lib, st = gpu_runtime.lib, gpu_runtime.stream()
    d = rdf.DeviceArray((1, 4, 4), np.uint16).fill(5)
    f = rdf.DeviceArray((1, 1, 9), np.float32).fill(0)
    assert lib.rdf_eval_forest(d.ptr, 1, 4, 4, f.ptr, 1, 1, 1, None, -1, d.ptr, 0, 1.0, st) == -1      # r = 0
    assert lib.rdf_eval_forest(d.ptr, 1, 4, 4, f.ptr, 1, 31, 1, None, -1, d.ptr, 1, 1.0, st) == -1     # depth 31
    assert lib.rdf_eval_forest(None, 1, 4, 4, f.ptr, 1, 1, 1, None, -1, d.ptr, 1, 1.0, st) == -2       # NULL
    assert lib.rdf_eval_forest(d.ptr, 1, 4, 4, f.ptr, 1, 1, 1, None, 2, d.ptr, 1, 1.0, st) == -2        # filter NULL
    assert lib.rdf_eval_forest(d.ptr, 70000, 1 << 15, 1, f.ptr, 1, 1, 1, None, -1, d.ptr, 1, 1.0, st) == -3
    assert lib.rdf_eval_forest(d.ptr, 0, 4, 4, f.ptr, 1, 1, 1, None, -1, d.ptr, 1, 1.0, st) == 0        # empty batch
    assert lib.rdf_eval_forest(d.ptr, 1, 4, 4, f.ptr, 1, 1, 1, None, -1, d.ptr, 8, 1.0, st) == 0        # r > dims


def test_config5_shape_bit_exact(rdf, evs, oracle):
    """BASELINE config 5's shape: 1280x720 frames, 8-tree depth-22 forest (1.875 GiB; packed tables 2.5 GiB).
    Two frames so that the oracle finishes in seconds."""
    synth = rdf.synth
    forest = synth.forest(8, 22, 4, "full")
    depth = synth.frames(["dense", "live"], 5000, 720, 1280)
    want = np.full(depth.shape, 65535, np.uint16)
    oracle.eval_forest(depth, forest, want)
    got = _gpu_forest(rdf, evs["packed"], depth, forest, 65535)
    assert np.array_equal(got, want), f"{(got != want).sum()} pixels differ"
    del forest


def test_hipgraph_capture_and_replay(rdf, evs, oracle, gpu_runtime):
    """The forest launch (dynamic tile queue with a self-resetting slot) can be captured into a hipGraph and
    replayed: same labels on every replay, also after the input frame changes in place."""
    import torch
    synth = rdf.synth
    forest_np = synth.forest(4, 10, 4, "trained")
    forest = rdf.DecisionForest.from_numpy(forest_np)
    frames = synth.frames(["live", "dense"], 60, 120, 160)
    depth = rdf.to_device(frames[0:1])
    labels = rdf.DeviceArray((1, 120, 160), np.uint16)
    ev = evs["packed"]
    forest.packed(1.0)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):          # warm-up on the capture stream (occupancy query, queue slot)
        labels.fill(65535)
        ev.get_labels_forest(forest, depth, labels)
    side.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        labels.fill(65535)
        ev.get_labels_forest(forest, depth, labels)
    for i in (0, 1, 0):
        depth.set(frames[i:i + 1])
        graph.replay()
        torch.cuda.synchronize()
        want = np.full((1, 120, 160), 65535, np.uint16)
        oracle.eval_forest(frames[i:i + 1], forest_np, want)
        assert np.array_equal(labels.get(), want), f"replay on frame {i}"


def test_fuzz_against_oracle(rdf, evs, oracle, gpu_runtime):
    """Seeded fuzz: 60 random (shape, forest, reduce, scale, filter, pre-fill, launch-geometry) combinations,
    each bit-exact against the C oracle on both paths."""
    rounds = int(os.environ.get("RDF_FUZZ_ROUNDS", "60"))     # a long soak: RDF_FUZZ_ROUNDS=2000
    rng = np.random.default_rng(int(os.environ.get("RDF_FUZZ_SEED", "20211003")))
    lib = gpu_runtime.lib
    try:
        for it in range(rounds):
            T, D, C = int(rng.integers(1, 10)), int(rng.integers(1, 12)), int(rng.integers(1, 20))
            n, h, w = int(rng.integers(1, 4)), int(rng.integers(1, 90)), int(rng.integers(1, 200))
            if rng.random() < 0.4:
                w = max(8, w & ~7)      # widths the 16-byte tile staging accepts
            r = int(rng.choice([1, 1, 2, 3, 7]))
            s = float(rng.choice([1.0, 0.5, 0.25, 1.5, 2.0]))
            forest = rdf.synth.forest(T, D, C, str(rng.choice(["full", "trained"])), first_tree=it)
            if rng.random() < 0.3:      # a few wild nodes: huge / tiny / non-finite numerators and thresholds
                k = (forest.shape[1] + 1) // 2
                with np.errstate(invalid="ignore", over="ignore"):      # (inf * 0 and overflow are the point here)
                    forest[:, :k, 0:5] *= rng.choice([1e-30, 1e30, 1e6, np.nan, np.inf], size=(T, k, 5)).astype(np.float32) \
                        * (rng.random((T, k, 5)) < 0.2) + (rng.random((T, k, 5)) >= 0.2)
            kinds = [str(k) for k in rng.choice(["dense", "live"], size=n)]
            depth = rdf.synth.frames(kinds, 3000 + it, h, w)
            depth[rng.random(depth.shape) < 0.02] = 0
            use_filter = rng.random() < 0.4
            filt = rng.integers(0, 3, size=(n, h // r, w // r)).astype(np.uint16) if use_filter else None
            prefill = int(rng.choice([65535, 0, 31337]))
            lib.rdf_set_block_threads(int(rng.choice([0, 256, 512, 1024])))
            lib.rdf_set_halo(int(rng.choice([-1, 0, 5, 16, 33, 48])))
            lib.rdf_set_lds_levels(int(rng.choice([-1, -1, 0, 2, 9])))
            lib.rdf_set_stage_vec(int(rng.choice([-1, -1, 0])))
            lib.rdf_set_group(int(rng.choice([0, 0, 1, 2, 3, 4])))
            lib.rdf_set_rows_per_wave(int(rng.choice([0, 1, 2, 4])))
            lib.rdf_set_scheduler(int(rng.choice([-1, 0, 1, 2])))
            lib.rdf_set_lds_budget_bytes(int(rng.choice([0, 1, 9000, 40000, 120000])))
            lib.rdf_set_tree_waves(int(rng.choice([-1, -1, 0, 1])))
            lib.rdf_set_last_level_table(int(rng.choice([-1, -1, 0])))
            lib.rdf_set_deep_from(int(rng.choice([-1, 0, 1, 4, 7, 10])))
            lib.rdf_set_fold(int(rng.choice([-1, -1, 0])))
            want = np.full((n, h // r, w // r), prefill, np.uint16)
            oracle.eval_forest(depth, forest, want, r, filt, 2 if use_filter else None, s)
            for path in ("packed", "direct"):
                got = _gpu_forest(rdf, evs[path], depth, forest, prefill, r, filt, 2 if use_filter else None, s)
                assert np.array_equal(got, want), f"iteration {it} ({path}): T{T} D{D} C{C} {n}x{h}x{w} r{r} s{s}"
    finally:
        lib.rdf_set_fold(-1)
        lib.rdf_set_block_threads(0)
        lib.rdf_set_halo(-1)
        lib.rdf_set_lds_levels(-1)
        lib.rdf_set_stage_vec(-1)
        lib.rdf_set_group(0)
        lib.rdf_set_rows_per_wave(0)
        lib.rdf_set_scheduler(-1)
        lib.rdf_set_lds_budget_bytes(0)
        lib.rdf_set_tree_waves(-1)
        lib.rdf_set_last_level_table(-1)
        lib.rdf_set_deep_from(-1)


@pytest.mark.parametrize("w,r", [(848, 1), (80, 1), (144, 1), (96, 1), (160, 1), (65, 1), (100, 1), (128, 1), (848, 2), (352, 2), (200, 3)])
def test_narrow_tiles_for_the_last_columns(w, r, rdf, evs, oracle, gpu_runtime):
    """Round 6: a label map that is 1 to 32 columns wider than a multiple of 64 gets narrow tiles for those columns (a wave =
    4 rows x 16 or 2 rows x 32 pixels) instead of a column of tiles whose waves run mostly empty -- 848 = 13 x 64 + 16 is every
    frame of the metric.  Same labels as the oracle's on every route a launch can take: packed and reference layout, one frame
    (small launch: a wave per tree, or four trees per lane) and a batch that fills the chip (512-thread workgroups), a filter
    (pixel lists), deep blocks, the fused fill; heights that are no multiple of the narrow tile's rows; and with the knob off."""
    lib = gpu_runtime.lib
    h = 101 if w < 400 else 480
    forest = rdf.synth.forest(4, 9, 4, "trained", 40 + w)
    n_big = 24 if w >= 352 else 400
    depth = rdf.synth.mixed_batch(n_big, 1200 + w, h, w)
    lh, lw = h // r, w // r
    rem = lw % 64
    assert (0 < rem <= 32 and lw > 64) == ((w, r) in ((848, 1), (80, 1), (144, 1), (96, 1), (160, 1), (65, 1), (200, 3))), (lw, rem)
    want = np.full((n_big, lh, lw), 65535, np.uint16)
    oracle.eval_forest(depth, forest, want, r)
    filt = (np.arange(n_big * lh * lw).reshape(n_big, lh, lw) % 3).astype(np.uint16)
    want_f = np.full((n_big, lh, lw), 7, np.uint16)
    oracle.eval_forest(depth, forest, want_f, r, filt, 1)
    f = rdf.DecisionForest.from_numpy(forest)
    d = rdf.to_device(depth)
    fd = rdf.to_device(filt)
    try:
        for fold in (-1, 0):
            lib.rdf_set_fold(fold)
            for path in ("packed", "direct"):
                ev = evs[path]
                for n in (1, n_big):                                  # a small launch and one that fills the chip
                    for tw in ((-1, 0) if n == 1 and path == "packed" else (-1,)):
                        lib.rdf_set_tree_waves(tw)
                        out = rdf.DeviceArray((n, lh, lw), np.uint16).fill(65535)
                        ev.get_labels_forest(f, d[0:n], out, r)
                        got = out.get()
                        assert np.array_equal(got, want[0:n]), (fold, path, n, tw, int((got != want[0:n]).sum()))
                    out = rdf.DeviceArray((n, lh, lw), np.uint16).fill(7)
                    ev.get_labels_forest(f, d[0:n], out, r, fd[0:n], 1)
                    got = out.get()
                    assert np.array_equal(got, want_f[0:n]), (fold, path, n, "filter", int((got != want_f[0:n]).sum()))
            out = rdf.DeviceArray((n_big, lh, lw), np.uint16).fill(12345)
            evs["packed"].get_labels_forest_filled(f, d, out, r)
            assert np.array_equal(out.get(), want), (fold, "filled")
            lib.rdf_set_deep_from(4)
            for n in (1, n_big):
                out = rdf.DeviceArray((n, lh, lw), np.uint16).fill(65535)
                evs["packed"].get_labels_forest(f, d[0:n], out, r)
                assert np.array_equal(out.get(), want[0:n]), (fold, "deep", n)
            lib.rdf_set_deep_from(-1)
    finally:
        lib.rdf_set_fold(-1)
        lib.rdf_set_tree_waves(-1)
        lib.rdf_set_deep_from(-1)


@pytest.mark.parametrize("trees,classes,r,topology", [(4, 4, 2, "full"), (4, 4, 1, "trained"), (3, 9, 2, "trained"), (2, 20, 1, "full"),
                                                      (4, 3, 3, "trained"), (2, 5, 2, "trained")])
def test_tree_waves_give_the_oracles_labels(trees, classes, r, topology, rdf, evs, oracle, gpu_runtime):
    """Small packed launches of 2-4 trees run one WAVE per tree and pixel row and meet in LDS (k_eval_forest<..., TW>;
    rdf_set_tree_waves, default on).  The same labels as the oracle and as the four-trees-in-a-lane kernel: live and dense
    frames, odd frame sizes (partial tiles, idle waves for three trees), class counts beyond one register chunk, pixels
    that reach no leaf (pre-fill kept), every pre-fill value, and a launch replayed from a captured graph."""
    import torch
    lib = gpu_runtime.lib
    synth = rdf.synth
    forest = synth.forest(trees, 11, classes, topology, first_tree=300 + trees)
    if topology == "trained":
        forest[:, -(forest.shape[1] + 1) // 2:, 5:7] = -1.0        # last level says "continue": those pixels get no label
    for n, h, w, kinds in ((1, 240, 424, ["live"]), (1, 97, 131, ["dense"]), (2, 64, 200, ["dense", "live"])):
        depth = synth.frames(kinds, 5100 + h, h, w)
        depth[::7, ::5] = 0
        for prefill in (65535, 0):
            want = np.full((n, h // r, w // r), prefill, np.uint16)
            oracle.eval_forest(depth, forest, want, r, None, None, 1.0)
            got = {}
            try:
                for mode in (1, 0):
                    lib.rdf_set_tree_waves(mode)
                    got[mode] = _gpu_forest(rdf, evs["packed"], depth, forest, prefill, r, None, None, 1.0)
            finally:
                lib.rdf_set_tree_waves(-1)
            assert np.array_equal(got[1], want), (trees, classes, r, topology, h, w, prefill, "tree waves")
            assert np.array_equal(got[0], want), (trees, classes, r, topology, h, w, prefill, "trees in a lane")
    # replayed from a graph, on changing frames
    F = rdf.DecisionForest.from_numpy(forest)
    F.packed(1.0)
    d_dev = rdf.to_device(synth.frames(["live"], 5300, 240, 424))
    out = rdf.DeviceArray((1, 240 // r, 424 // r), np.uint16)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        out.fill(65535)
        evs["packed"].get_labels_forest(F, d_dev, out, labels_reduce=r)
    side.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        out.fill(65535)
        evs["packed"].get_labels_forest(F, d_dev, out, labels_reduce=r)
    for it in range(3):
        frame = synth.frames(["live" if it % 2 else "dense"], 5400 + it, 240, 424)
        d_dev.set(frame)
        graph.replay()
        torch.cuda.synchronize()
        want = np.full((1, 240 // r, 424 // r), 65535, np.uint16)
        oracle.eval_forest(frame, forest, want, r, None, None, 1.0)
        assert np.array_equal(out.get(), want), it


def test_two_streams_do_not_share_queue_state(rdf, evs, oracle, gpu_runtime):
    """Launches on two streams run concurrently with their own tile-queue slots."""
    import torch
    synth = rdf.synth
    fa, fb = synth.forest(4, 12, 4, "trained", 1), synth.forest(3, 11, 5, "full", 9)
    da, db = synth.frames(["dense"] * 6, 800, 240, 424), synth.frames(["live"] * 6, 900, 200, 300)
    wa, wb = np.full(da.shape, 65535, np.uint16), np.full(db.shape, 65535, np.uint16)
    oracle.eval_forest(da, fa, wa)
    oracle.eval_forest(db, fb, wb)
    Fa, Fb = rdf.DecisionForest.from_numpy(fa), rdf.DecisionForest.from_numpy(fb)
    Da, Db = rdf.to_device(da), rdf.to_device(db)
    La, Lb = rdf.DeviceArray(da.shape, np.uint16), rdf.DeviceArray(db.shape, np.uint16)
    Fa.packed(1.0), Fb.packed(1.0)
    torch.cuda.synchronize()
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    ev = evs["packed"]
    for _ in range(5):
        with torch.cuda.stream(sa):
            La.fill(65535)
            ev.get_labels_forest(Fa, Da, La)
        with torch.cuda.stream(sb):
            Lb.fill(65535)
            ev.get_labels_forest(Fb, Db, Lb)
    torch.cuda.synchronize()
    assert np.array_equal(La.get(), wa) and np.array_equal(Lb.get(), wb)


def test_largest_batch_one_call_addresses(rdf, evs, oracle):
    """The C ABI addresses a batch with 32-bit pixel offsets (n_img*W*H < 2^31).  Run it at that limit: 5275
    frames of 848x480 = 2 147 136 000 pixels (4.3 GB of depth, 4.3 GB of labels), a small forest, and check the
    first, a middle and the last frames against the oracle -- an offset that wrapped would show there."""
    import torch
    n, h, w = 5275, 480, 848
    assert n * h * w < 2 ** 31 <= (n + 1) * h * w
    free, _ = torch.cuda.mem_get_info()
    if free < 24 * 2 ** 30:
        pytest.skip("needs 24 GB of free HBM")
    forest = rdf.synth.forest(3, 5, 4, "trained", first_tree=90)
    depth = rdf.DeviceArray((n, h, w), np.uint16)
    # every frame = one of 7 synthetic frames, rolled by a frame-dependent amount: cheap to rebuild on the host
    base = rdf.synth.frames(["dense", "live", "dense", "live", "live", "dense", "live"], 300, h, w)
    base_t = torch.from_numpy(base.view(np.int16)).cuda()
    d_t = depth.torch_bytes().view(torch.int16).view(n, h, w)
    for i in range(n):
        d_t[i] = torch.roll(base_t[i % 7], shifts=(i * 13) % w, dims=1)
    depth.mark_dirty()
    labels = rdf.DeviceArray((n, h, w), np.uint16).fill(65535)
    f = rdf.DecisionForest.from_numpy(forest)
    evs["packed"].get_labels_forest(f, depth, labels)
    for i in (0, 1, 2637, n - 2, n - 1):
        host = np.roll(base[i % 7], (i * 13) % w, axis=1)[None]
        want = np.full((1, h, w), 65535, np.uint16)
        oracle.eval_forest(host, forest, want)
        got = labels[i:i + 1].get()
        assert np.array_equal(got, want), f"frame {i}: {(got != want).sum()} pixels differ"
    # nothing was written past the batch's own frames: every frame's untouched pixels are still the pre-fill
    l_t = labels.torch_bytes().view(torch.int16).view(n, h, w)
    invalid = (d_t == 0) | (d_t == -1)
    assert not bool(torch.logical_and(l_t != -1, invalid).any())
    assert not bool(torch.logical_and(l_t == -1, ~invalid).any())   # and every valid pixel did get a label
    del depth, labels, d_t, l_t, invalid
    torch.cuda.empty_cache()


def test_cu_masked_stream_gives_the_same_labels(rdf, evs, oracle, gpu_runtime):
    """The compute stream bench.py uses next to RCCL (one CU per shader engine left out): same results, and the
    argument checks of the stream helper."""
    import ctypes
    import torch
    lib = gpu_runtime.lib
    h = ctypes.c_void_p()
    assert lib.rdf_stream_create_with_reserved_cus(ctypes.byref(h), 0) == -1
    assert lib.rdf_stream_create_with_reserved_cus(ctypes.byref(h), 100000) == -1
    assert lib.rdf_stream_create_with_reserved_cus(None, 32) != 0
    assert lib.rdf_stream_create_with_reserved_cus(ctypes.byref(h), 32) == 0
    try:
        forest_np = rdf.synth.forest(4, 12, 4, "trained", first_tree=33)
        depth = rdf.synth.mixed_batch(6, first_idx=4321, h=240, w=424)
        want = np.full(depth.shape, 65535, np.uint16)
        oracle.eval_forest(depth, forest_np, want)
        forest = rdf.DecisionForest.from_numpy(forest_np)
        d_dev = rdf.to_device(depth)
        out = rdf.DeviceArray(depth.shape, np.uint16).fill(65535)
        torch.cuda.synchronize()
        with torch.cuda.stream(torch.cuda.ExternalStream(h.value)):
            for _ in range(2):
                evs["packed"].get_labels_forest(forest, d_dev, out)
            got = out.get()
        assert np.array_equal(got, want)
    finally:
        torch.cuda.synchronize()
        assert lib.rdf_stream_destroy(h) == 0


def test_cu_masked_stream_sizes_the_grid_on_the_reference_layout_path_too(rdf, evs, oracle, gpu_runtime):
    """The unpacked entry point on a CU-masked stream: the persistent grid must be sized from the stream's CUs (round 1
    passed the device's CU count on this one path) and the labels are the oracle's."""
    import ctypes
    import torch
    lib = gpu_runtime.lib
    h = ctypes.c_void_p()
    assert lib.rdf_stream_create_with_reserved_cus(ctypes.byref(h), 64) == 0
    try:
        forest_np = rdf.synth.forest(4, 11, 4, "trained", first_tree=35)
        depth = rdf.synth.mixed_batch(8, first_idx=4400, h=240, w=424)
        want = np.full(depth.shape, 65535, np.uint16)
        oracle.eval_forest(depth, forest_np, want)
        forest = rdf.DecisionForest.from_numpy(forest_np)
        d_dev = rdf.to_device(depth)
        out = rdf.DeviceArray(depth.shape, np.uint16).fill(65535)
        torch.cuda.synchronize()
        with torch.cuda.stream(torch.cuda.ExternalStream(h.value)):
            evs["direct"].get_labels_forest(forest, d_dev, out)
            got = out.get()
        assert np.array_equal(got, want)
    finally:
        torch.cuda.synchronize()
        assert lib.rdf_stream_destroy(h) == 0


@pytest.mark.parametrize("pattern", ["first_ses", "every_8th", "low_word", "one_cu"])
def test_xcd_queues_are_drained_when_some_xcds_run_no_workgroup(pattern, rdf, evs, oracle, gpu_runtime):
    """The tile queues are per XCD (a workgroup drains its own XCD's range and helps two neighbours); a CU mask may leave
    whole XCDs without a workgroup of the launch, whose ranges the workgroups that go round all queues must drain.
    Streams with CU masks of several shapes (whatever the bit -> XCD numbering is, some of them empty some XCDs), a
    launch of more than 64 workgroups so that the per-XCD queues are in use: the oracle's labels, twice in a row (the
    slot resets itself)."""
    import ctypes
    import torch
    lib = gpu_runtime.lib
    hip = ctypes.CDLL("libamdhip64.so")
    n_cu = torch.cuda.get_device_properties(0).multi_processor_count
    bits = {"first_ses": [i for i in range(n_cu) if i % 32 < 4], "every_8th": [i for i in range(n_cu) if i % 8 == 0],
            "low_word": list(range(32)), "one_cu": [5]}[pattern]
    words = (ctypes.c_uint32 * 8)()
    for b in bits:
        words[b // 32] |= 1 << (b % 32)
    h = ctypes.c_void_p()
    assert hip.hipExtStreamCreateWithCUMask(ctypes.byref(h), 8, words) == 0
    try:
        forest_np = rdf.synth.forest(4, 10, 4, "trained", first_tree=61)
        depth = rdf.synth.mixed_batch(24 if pattern != "one_cu" else 3, first_idx=4700, h=240, w=424)
        want = np.full(depth.shape, 65535, np.uint16)
        oracle.eval_forest(depth, forest_np, want)
        forest = rdf.DecisionForest.from_numpy(forest_np)
        d_dev = rdf.to_device(depth)
        out = rdf.DeviceArray(depth.shape, np.uint16)
        torch.cuda.synchronize()
        with torch.cuda.stream(torch.cuda.ExternalStream(h.value)):
            for name in ("packed", "direct"):
                for _ in range(2):
                    out.fill(65535)
                    evs[name].get_labels_forest(forest, d_dev, out)
                    assert np.array_equal(out.get(), want), (pattern, name)
    finally:
        torch.cuda.synchronize()
        assert lib.rdf_stream_destroy(h) == 0


def test_destroyed_streams_give_their_queue_slots_back(rdf, evs, oracle, gpu_runtime):
    """A device has 128 tile-queue slots for streams; rdf_stream_destroy returns a stream's slot, so a program that
    creates and destroys more streams than that keeps the dynamic queue (round 1 leaked the slot and fell back to
    static tiles for good after that many streams)."""
    import ctypes
    import torch
    lib = gpu_runtime.lib
    forest_np = rdf.synth.forest(4, 9, 4, "trained", first_tree=36)
    depth = rdf.synth.mixed_batch(2, first_idx=4500, h=120, w=200)
    want = np.full(depth.shape, 65535, np.uint16)
    oracle.eval_forest(depth, forest_np, want)
    forest = rdf.DecisionForest.from_numpy(forest_np)
    d_dev = rdf.to_device(depth)
    out = rdf.DeviceArray(depth.shape, np.uint16)
    used, graph = ctypes.c_int(), ctypes.c_int()
    assert lib.rdf_debug_sched_slots(ctypes.byref(used), ctypes.byref(graph)) == 0
    before = used.value
    for i in range(300):
        h = ctypes.c_void_p()
        assert lib.rdf_stream_create_with_reserved_cus(ctypes.byref(h), 32) == 0
        with torch.cuda.stream(torch.cuda.ExternalStream(h.value)):
            out.fill(65535)
            evs["packed"].get_labels_forest(forest, d_dev, out)
        assert lib.rdf_stream_destroy(h) == 0            # waits for the stream's work
        if i % 97 == 0:
            assert np.array_equal(out.get(), want), i
    assert lib.rdf_debug_sched_slots(ctypes.byref(used), ctypes.byref(graph)) == 0
    assert used.value <= before + 1, (before, used.value)
    assert np.array_equal(out.get(), want)


def test_captured_launch_has_its_own_queue_slot_and_replays_next_to_direct_launches(rdf, evs, oracle, gpu_runtime):
    """A forest launch recorded into a hipGraph gets a tile-queue slot of its own: the graph is replayed on ANOTHER stream
    while direct launches run on the stream it was captured on (round 1 keyed both by the capture stream's handle: two
    concurrent launches then stole each other's tiles).  Every replay and every direct launch gives the oracle's labels."""
    import ctypes
    import torch
    lib = gpu_runtime.lib
    synth = rdf.synth
    fa, fb = synth.forest(4, 11, 4, "trained", 3), synth.forest(4, 10, 4, "full", 7)
    da, db = synth.frames(["live", "dense", "live"], 820, 240, 424), synth.frames(["dense", "live"], 830, 200, 300)
    wa, wb = np.full(da.shape, 65535, np.uint16), np.full(db.shape, 65535, np.uint16)
    oracle.eval_forest(da, fa, wa)
    oracle.eval_forest(db, fb, wb)
    Fa, Fb = rdf.DecisionForest.from_numpy(fa), rdf.DecisionForest.from_numpy(fb)
    Da, Db = rdf.to_device(da), rdf.to_device(db)
    La, Lb = rdf.DeviceArray(da.shape, np.uint16), rdf.DeviceArray(db.shape, np.uint16)
    Fa.packed(1.0), Fb.packed(1.0)
    ev = evs["packed"]
    cap, other = torch.cuda.Stream(), torch.cuda.Stream()
    with torch.cuda.stream(cap):
        La.fill(65535)
        ev.get_labels_forest(Fa, Da, La)
    cap.synchronize()
    used, g0, g1 = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    lib.rdf_debug_sched_slots(ctypes.byref(used), ctypes.byref(g0))
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=cap):
        La.fill(65535)
        ev.get_labels_forest(Fa, Da, La)
    lib.rdf_debug_sched_slots(ctypes.byref(used), ctypes.byref(g1))
    assert g1.value == g0.value + 1
    for it in range(20):
        with torch.cuda.stream(other):
            graph.replay()                              # on a stream the graph was not captured on
        with torch.cuda.stream(cap):
            Lb.fill(65535)
            ev.get_labels_forest(Fb, Db, Lb)            # direct launch on the capture stream, concurrently
        torch.cuda.synchronize()
        assert np.array_equal(La.get(), wa), it
        assert np.array_equal(Lb.get(), wb), it


@pytest.mark.parametrize("one_launch,tree_waves", [(1, 1), (1, 0), (0, 1)])
def test_layers_in_one_launch_equal_the_filtered_sequence(one_launch, tree_waves, rdf, gpu_runtime, oracle):
    """A single-frame run of a packed stack evaluates its layers unfiltered in ONE launch (workgroup b takes layer b % n) and
    filters in the composite kernel (rdf_set_layers_one_launch, default on for small launches); with it off the layers run
    one after the other, filtered.  Same per-layer label images and composite either way: the oracle's chain -- on
    changing frames, replayed from a captured graph, and with a layer whose filter class no pixel has (layers 1 and 2
    then come out all-65535).  With tree waves on (the default) the one launch runs every tree of a layer in a wave of its own
    (all three forests have 3-4 trees); off, four trees in a lane."""
    import torch
    synth = rdf.synth
    lib = gpu_runtime.lib
    h, w, r = 240, 424, 2
    forests = [synth.forest(4, 9, 4, "trained", 200), synth.forest(3, 10, 5, "trained", 210), synth.forest(4, 8, 3, "trained", 220)]
    conditions = [[0, 1], [1, 4], [0, 2], [0, 3], [0, 4], [1, 9], [0, 5], [0, 6], [0, 7], [0, 8], [0, 9], [0, 10]]
    cfg = {"layers": [{"model": rdf.DecisionForest.from_numpy(forests[0])},
                      {"model": rdf.DecisionForest.from_numpy(forests[1]), "filter_model": 0, "filter_model_class": 2},
                      {"model": rdf.DecisionForest.from_numpy(forests[2]), "filter_model": 1, "filter_model_class": 2}],
           "conditions": conditions, "label_colors": [[i, i, i, 255] for i in range(10)]}
    lib.rdf_set_layers_one_launch(one_launch)
    lib.rdf_set_tree_waves(tree_waves)
    try:
        lf = rdf.LayeredDecisionForest(cfg, (h, w), r)
        dbuf, lbuf = rdf.GpuBuffer((h, w), np.uint16), rdf.GpuBuffer((h // r, w // r), np.uint16)
        frames = synth.frames(["live", "dense", "live"], 930, h, w)

        def want_for(frame, cls1):
            shape = (1, h // r, w // r)
            l0, l1, l2, comp = (np.full(shape, 65535, np.uint16) for _ in range(4))
            oracle.eval_forest(frame[None], forests[0], l0, r, None, None, 0.5)
            oracle.eval_forest(frame[None], forests[1], l1, r, l0, cls1, 0.5)
            oracle.eval_forest(frame[None], forests[2], l2, r, l1, 2, 0.5)
            oracle.composite([l0[0], l1[0], l2[0]], np.array(conditions, np.int32), comp)
            return l0[0], l1[0], l2[0], comp[0]

        for i in range(3):
            dbuf.cu().set(frames[i])
            lbuf.cu().fill(7)
            lf.run(dbuf, lbuf, 0.5)
            l0, l1, l2, comp = want_for(frames[i], 2)
            assert np.array_equal(lf.label_images[0].cu().get(), l0), i
            assert np.array_equal(lf.label_images[1].cu().get(), l1), i
            assert np.array_equal(lf.label_images[2].cu().get(), l2), i
            assert np.array_equal(lbuf.cu().get(), comp), i
            assert (l1 != 65535).sum() > 50
        # captured as a graph on a side stream, replayed on changing frames
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            lf.run(dbuf, lbuf, 0.5)
        side.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            lf.run(dbuf, lbuf, 0.5)
        for i in (1, 0, 2):
            dbuf.cu().set(frames[i])
            graph.replay()
            torch.cuda.synchronize()
            l0, l1, l2, comp = want_for(frames[i], 2)
            assert np.array_equal(lbuf.cu().get(), comp) and np.array_equal(lf.label_images[2].cu().get(), l2), i
        # a filter class that no pixel of layer 0 has: layers 1 and 2 must come out all-65535
        cfg["layers"][1]["filter_model_class"] = 9
        lf9 = rdf.LayeredDecisionForest(cfg, (h, w), r)
        dbuf.cu().set(frames[1])
        lf9.run(dbuf, lbuf, 0.5)
        l0, l1, l2, comp = want_for(frames[1], 9)
        assert (l1 == 65535).all() and (l2 == 65535).all()
        assert np.array_equal(lf9.label_images[1].cu().get(), l1) and np.array_equal(lf9.label_images[2].cu().get(), l2)
        assert np.array_equal(lbuf.cu().get(), comp)
    finally:
        torch.cuda.synchronize()
        lib.rdf_set_layers_one_launch(-1)
        lib.rdf_set_tree_waves(-1)


@pytest.mark.parametrize("tree_waves", [0, 1])
def test_layered_run_with_a_layer_whose_table_walks_deep_blocks(tree_waves, rdf, gpu_runtime, oracle):
    """A stack whose first layer's packed table carries a deep-level choice: the layers-in-one-launch kernel has no block slabs, so
    such a layer takes a launch of its own (tree waves, when on, come first and walk no deep blocks) -- the per-layer label
    images and the composite are the oracle's chain either way."""
    synth = rdf.synth
    lib = gpu_runtime.lib
    h, w, r, s_ = 240, 424, 2, 0.5
    forests = [synth.forest(4, 11, 4, "balanced", 300, calib=synth.calibration_frames(3, 120, 212)), synth.forest(3, 9, 4, "trained", 310)]
    conditions = [[0, 1], [0, 2], [1, 3], [0, 3], [0, 4], [0, 5], [0, 6]]
    models = [rdf.DecisionForest.from_numpy(f) for f in forests]
    cfg = {"layers": [{"model": models[0]}, {"model": models[1], "filter_model": 0, "filter_model_class": 3}],
           "conditions": conditions, "label_colors": [[i, i, i, 255] for i in range(6)]}
    assert lib.rdf_forest_set_deep_from(models[0].packed(s_).ptr, 6) == 0 and models[0].deep_from(s_) == 6
    lib.rdf_set_tree_waves(tree_waves)
    try:
        lf = rdf.LayeredDecisionForest(cfg, (h, w), r)
        dbuf, lbuf = rdf.GpuBuffer((h, w), np.uint16), rdf.GpuBuffer((h // r, w // r), np.uint16)
        for frame in synth.frames(["live", "dense"], 960, h, w):
            dbuf.cu().set(frame)
            lf.run(dbuf, lbuf, s_)
            shape = (1, h // r, w // r)
            l0, l1, comp = (np.full(shape, 65535, np.uint16) for _ in range(3))
            oracle.eval_forest(frame[None], forests[0], l0, r, None, None, s_)
            oracle.eval_forest(frame[None], forests[1], l1, r, l0, 3, s_)
            oracle.composite([l0[0], l1[0]], np.array(conditions, np.int32), comp)
            assert np.array_equal(lf.label_images[0].cu().get().reshape(shape), l0)
            assert np.array_equal(lf.label_images[1].cu().get().reshape(shape), l1)
            assert np.array_equal(lbuf.cu().get().reshape(shape), comp)
    finally:
        lib.rdf_set_tree_waves(-1)


@pytest.mark.parametrize("one_launch", [1, 0])
def test_filter_class_minus_one_means_no_filter(one_launch, rdf, gpu_runtime, oracle):
    """A layer that names a filter layer but the filter class -1 is evaluated everywhere (tree_eval.cu:81: the filter is
    only looked at when filter_class != -1) -- through the one-launch route (the composite kernel applies the filters
    there) as through filtered launches."""
    import torch
    synth = rdf.synth
    lib = gpu_runtime.lib
    h, w, r = 240, 424, 2
    forests = [synth.forest(3, 8, 4, "trained", 300), synth.forest(4, 9, 3, "trained", 310)]
    conditions = [[0, 1], [0, 2], [1, 4], [0, 3], [0, 4], [0, 5], [0, 6]]
    cfg = {"layers": [{"model": rdf.DecisionForest.from_numpy(forests[0])},
                      {"model": rdf.DecisionForest.from_numpy(forests[1]), "filter_model": 0, "filter_model_class": -1}],
           "conditions": conditions, "label_colors": [[i, i, i, 255] for i in range(6)]}
    lib.rdf_set_layers_one_launch(one_launch)
    try:
        lf = rdf.LayeredDecisionForest(cfg, (h, w), r)
        dbuf, lbuf = rdf.GpuBuffer((h, w), np.uint16), rdf.GpuBuffer((h // r, w // r), np.uint16)
        frame = synth.frames(["live"], 955, h, w)[0]
        dbuf.cu().set(frame)
        lf.run(dbuf, lbuf, 1.0)
        shape = (1, h // r, w // r)
        l0, l1, comp = (np.full(shape, 65535, np.uint16) for _ in range(3))
        oracle.eval_forest(frame[None], forests[0], l0, r, None, None, 1.0)
        oracle.eval_forest(frame[None], forests[1], l1, r, None, None, 1.0)     # unfiltered
        oracle.composite([l0[0], l1[0]], np.array(conditions, np.int32), comp)
        assert (l1 != 65535).sum() > 1000
        assert np.array_equal(lf.label_images[0].cu().get(), l0[0])
        assert np.array_equal(lf.label_images[1].cu().get(), l1[0])
        assert np.array_equal(lbuf.cu().get(), comp[0])
    finally:
        torch.cuda.synchronize()
        lib.rdf_set_layers_one_launch(-1)


@pytest.mark.parametrize("trees", [1, 2, 3, 6])
def test_small_forests_on_big_batches_take_narrow_walks(trees, rdf, gpu_runtime, oracle):
    """A batch that fills the chip runs 512-thread workgroups; forests of one, two, three or six trees then walk 1, 2 or 3
    trees in a lane (no idle slots), packed and from the reference layout: the oracle's labels."""
    synth = rdf.synth
    n, h, w = 40, 240, 424
    f_np = synth.forest(trees, 9, 4, "trained", 400 + trees)
    frames = synth.frames(["dense", "live"] * (n // 2), 970, h, w)
    want = np.full(frames.shape, 65535, np.uint16)
    oracle.eval_forest(frames, f_np, want)
    forest = rdf.DecisionForest.from_numpy(f_np)
    depth = rdf.to_device(frames)
    for packed in (True, False):
        labels = rdf.DeviceArray(frames.shape, np.uint16).fill(65535)
        rdf.DecisionTreeEvaluator(use_packed=packed).get_labels_forest(forest, depth, labels)
        assert np.array_equal(labels.get(), want), (trees, packed)


def test_layered_fuzz_against_oracle(rdf, gpu_runtime, oracle):
    """Seeded fuzz over layered stacks: 2-3 layers of random forests (1-9 trees, depth 1-11, 1-18 classes), random
    filter wiring (on any earlier layer or none), reduce, scale, frame size and conditions table; LayeredDecisionForest.run
    through all three routes -- layers in one launch, filtered launches one after the other, and the reference's
    step-by-step sequence -- against the oracle's chain, per-layer label images and composite."""
    rounds = int(os.environ.get("RDF_LAYERED_FUZZ_ROUNDS", "40"))
    rng = np.random.default_rng(int(os.environ.get("RDF_FUZZ_SEED", "20211003")) + 17)
    lib = gpu_runtime.lib
    try:
        for it in range(rounds):
            n_layers = int(rng.integers(2, 4))
            h, w = int(rng.integers(8, 140)), int(rng.integers(8, 260))
            if rng.random() < 0.4:
                w = max(8, w & ~7)
            r = int(rng.choice([1, 1, 2, 3]))
            if h // r == 0 or w // r == 0:
                continue
            s = float(rng.choice([1.0, 0.5, 1.5]))
            forests, layers, n_classes = [], [], []
            for i in range(n_layers):
                T, D, C = int(rng.integers(1, 10)), int(rng.integers(1, 12)), int(rng.integers(1, 19))
                if it % 2:
                    T = int(rng.integers(2, 5))       # stacks whose layers can all run as tree waves
                f = rdf.synth.forest(T, D, C, str(rng.choice(["full", "trained"])), first_tree=1000 + 10 * it + i)
                forests.append(f)
                n_classes.append(C)
                l = {"model": rdf.DecisionForest.from_numpy(f)}
                if i > 0 and rng.random() < 0.8:
                    l["filter_model"] = int(rng.integers(0, i))
                    l["filter_model_class"] = int(rng.integers(0, n_classes[l["filter_model"]] + 1))
                layers.append(l)
            # a conditions table that terminates on every path: layer i's classes either name a composite id or go on
            cond, n_ids = [], 0
            offs = [0]
            for i in range(n_layers):
                for c in range(1, n_classes[i] + 1):
                    if i + 1 < n_layers and rng.random() < 0.4:
                        cond.append([1, -1])          # patched below with the next layer's offset
                    else:
                        n_ids += 1
                        cond.append([0, n_ids])
                offs.append(len(cond))
            row = 0
            for i in range(n_layers):
                for c in range(n_classes[i]):
                    if cond[row][0] == 1:
                        cond[row][1] = offs[i + 1]
                    row += 1
            if n_ids == 0:
                cond[0] = [0, 1]
                n_ids = 1
            cfg = {"layers": layers, "conditions": cond, "label_colors": [[i % 256, 0, 0, 255] for i in range(n_ids)]}
            kind = str(rng.choice(["dense", "live"]))
            frame = rdf.synth.frames([kind], 7000 + it, h, w)[0]
            frame[rng.random(frame.shape) < 0.03] = 0
            shape = (1, h // r, w // r)
            want = [np.full(shape, 65535, np.uint16) for _ in range(n_layers)]
            for i in range(n_layers):
                fm = layers[i].get("filter_model")
                oracle.eval_forest(frame[None], forests[i], want[i], r, want[fm] if fm is not None else None,
                                   layers[i].get("filter_model_class"), s)
            comp = np.full(shape, 65535, np.uint16)
            oracle.composite([x[0] for x in want], np.array(cond, np.int32), comp)
            dbuf, lbuf = rdf.GpuBuffer((h, w), np.uint16), rdf.GpuBuffer((h // r, w // r), np.uint16)
            dbuf.cu().set(frame)
            for route in ("one launch", "one launch, trees in a lane", "filtered launches", "step by step"):
                lib.rdf_set_layers_one_launch(0 if route == "filtered launches" else -1)
                lib.rdf_set_tree_waves(0 if route == "one launch, trees in a lane" else -1)
                lf = rdf.LayeredDecisionForest(cfg, (h, w), r, fused=route != "step by step")
                lbuf.cu().fill(3)
                lf.run(dbuf, lbuf, s)
                for i in range(n_layers):
                    got = lf.label_images[i].cu().get()
                    assert np.array_equal(got, want[i][0]), f"iteration {it}, {route}, layer {i}: {(got != want[i][0]).sum()} pixels"
                assert np.array_equal(lbuf.cu().get(), comp[0]), f"iteration {it}, {route}: composite"
    finally:
        lib.rdf_set_layers_one_launch(-1)
        lib.rdf_set_tree_waves(-1)


@pytest.mark.parametrize("r,s", [(1, 1.0), (3, 0.5)])
def test_three_layer_stack_matches_oracle(r, s, rdf, gpu_runtime, oracle):
    """A 3-layer stack (layer 1 filtered on a class of layer 0, layer 2 on a class of layer 1) with a conditions
    table that chains through all three: one-call path and step-by-step path against the oracle's chain."""
    synth = rdf.synth
    h, w = 96, 150
    forests = [synth.forest(3, 7, 3, "trained", 100), synth.forest(2, 8, 4, "trained", 110), synth.forest(4, 6, 3, "trained", 120)]
    # layer0: 1 -> pixel 1, 2 -> go on (offset 3), 3 -> pixel 2; layer1 (offset 3): 1 -> 3, 2 -> 4, 3 -> go on (offset 7),
    # 4 -> 5; layer2 (offset 7): 1 -> 6, 2 -> 7, 3 -> 8
    conditions = [[0, 1], [1, 3], [0, 2], [0, 3], [0, 4], [1, 7], [0, 5], [0, 6], [0, 7], [0, 8]]
    cfg = {"layers": [{"model": rdf.DecisionForest.from_numpy(forests[0])},
                      {"model": rdf.DecisionForest.from_numpy(forests[1]), "filter_model": 0, "filter_model_class": 2},
                      {"model": rdf.DecisionForest.from_numpy(forests[2]), "filter_model": 1, "filter_model_class": 3}],
           "conditions": conditions, "label_colors": [[i, i, i, 255] for i in range(8)]}
    lf = rdf.LayeredDecisionForest(cfg, (h, w), r)
    assert lf.num_layered_classes == 8
    depth = synth.frames(["dense"], 555, h, w)
    depth[0, :10, :] = 0
    depth[0, 40:50, 60:90] = 65535
    shape = (1, h // r, w // r)
    l0, l1, l2, comp = (np.full(shape, 65535, np.uint16) for _ in range(4))
    oracle.eval_forest(depth, forests[0], l0, r, None, None, s)
    oracle.eval_forest(depth, forests[1], l1, r, l0, 2, s)
    oracle.eval_forest(depth, forests[2], l2, r, l1, 3, s)
    oracle.composite([l0[0], l1[0], l2[0]], np.array(conditions, np.int32), comp)
    assert len(np.unique(comp)) >= 5, "the example should exercise several branches of the table"
    dbuf, lbuf = rdf.GpuBuffer((h, w), np.uint16), rdf.GpuBuffer((h // r, w // r), np.uint16)
    dbuf.cu().set(depth[0])
    for fused in (True, False):
        lf.fused = fused
        lbuf.cu().fill(4242)
        lf.run(dbuf, lbuf, s)
        for got, want in zip(lf.label_images, (l0, l1, l2)):
            assert np.array_equal(got.cu().get(), want[0]), f"fused={fused}"
        assert np.array_equal(lbuf.cu().get(), comp[0]), f"fused={fused}"
    assert lf.eval.composite_bad_pixels() == 0


@pytest.mark.parametrize("spoil", ["none", "continue_flag", "huge_numerator", "nan_child"])
@pytest.mark.parametrize("trees,levels,classes,topology", [(4, 9, 4, "full"), (4, 10, 4, "trained"), (3, 9, 3, "trained"), (8, 9, 2, "full"),
                                                          (1, 8, 1, "full"), (2, 2, 4, "full")])
def test_last_level_table_is_taken_only_when_every_deepest_node_is_an_ordinary_one(trees, levels, classes, topology, spoil,
                                                                                    rdf, gpu_runtime, oracle):
    """Packed forests of up to four classes hold the nodes of level D-1 with both their leaf PDFs in 64-byte records
    (include/rdf_hip.h, rdf_forest_pack).  The table's trailer counts the records it cannot serve -- a child flag that says
    "continue" (tree_eval.cu:101-102 with -1 at the last level: no contribution), a numerator for the IEEE divide -- and one
    such record sends the whole forest down the general path; so does, by default, a forest that uses less than half of its
    deepest level.  With the knob forced on, off and left alone: the oracle's labels, on a batch that fills the chip
    (512-thread workgroups) and on a small one."""
    synth = rdf.synth
    rng = np.random.default_rng(5 + trees * 100 + levels)
    f_np = synth.forest(trees, levels, classes, topology, 700 + trees)
    first = (1 << (levels - 1)) - 1                     # 0-based index of the first node of level D-1
    f_np[:, first:, 7:] = rng.random((trees, first + 1, 2 * classes), dtype=np.float32)     # left and right PDFs tell sides apart
    k, node = trees - 1, first + int(rng.integers(0, first + 1))
    if spoil == "continue_flag":
        f_np[k, node, 5] = -1.0
    elif spoil == "huge_numerator":
        f_np[k, node, 2] = 3.0e7
    elif spoil == "nan_child":                          # NaN is not in [-1, 0): a leaf like any other
        f_np[k, node, 6] = np.nan
    lib = gpu_runtime.lib
    forest = rdf.DecisionForest.from_numpy(f_np)
    packed = forest.packed(1.0)
    gpu_runtime.synchronize()
    slots = trees << levels
    trailer = slots * (16 + 32) + (trees << (levels - 1)) * 64
    assert packed.nbytes >= trailer + 64           # (the deep blocks follow, 128-byte aligned; the info block is last)
    unusable, in_use = (int(x) for x in packed.get()[trailer:trailer + 8].view(np.uint32))
    # the deep blocks' trailer says the same in its own words: 1 + the deepest level with a node that needs the exact record,
    # and the nodes of the last level that are not plain two-leaf nodes
    if levels >= 5:
        deep_at = (trailer + 64 + 127) & ~127
        r0 = levels - 2
        deep_lines = trees * (((1 << r0) - (1 << (r0 % 3))) // 7) + (trees << r0)
        assert packed.nbytes == deep_at + (deep_lines + 2) * 128 + 128
        exact_below, last_bad = (int(x) for x in packed.get()[deep_at + (deep_lines + 1) * 128:][:8].view(np.uint32))
        assert exact_below == (levels if spoil == "huge_numerator" else 0) and last_bad == unusable
    else:
        assert packed.nbytes == ((trailer + 64 + 127) & ~127) + 128
    # the info block: nodes that need the exact numerators (they are read from the caller's forest), the scale, the mark
    info = packed.get()[packed.nbytes - 128:][:12]
    assert int(info[0:4].view(np.uint32)[0]) == (1 if spoil == "huge_numerator" else 0)
    assert float(info[4:8].view(np.float32)[0]) == 1.0 and int(info[8:12].view(np.uint32)[0]) == 0x52444635
    assert unusable == (1 if spoil in ("continue_flag", "huge_numerator") else 0)
    # word 1: the deepest nodes some parent continues to (the default takes the table from half of the level on)
    to_child = f_np[:, first // 2:first, 5:7].reshape(trees, -1)
    assert in_use == int(((to_child >= -1) & (to_child < 0)).sum())
    for n, h, w in ((40, 240, 424), (2, 70, 90)):
        frames = synth.frames(["dense", "live"] * (n // 2), 31 + n, h, w)
        want = np.full(frames.shape, 65535, np.uint16)
        oracle.eval_forest(frames, f_np, want)
        depth = rdf.to_device(frames)
        for knob in (1, 0, -1):
            lib.rdf_set_last_level_table(knob)
            try:
                labels = rdf.DeviceArray(frames.shape, np.uint16).fill(65535)
                rdf.DecisionTreeEvaluator(use_packed=True).get_labels_forest(forest, depth, labels)
                assert np.array_equal(labels.get(), want), (n, knob, int((labels.get() != want).sum()))
            finally:
                lib.rdf_set_last_level_table(-1)


def test_last_level_table_fuzz(rdf, evs, oracle, gpu_runtime):
    """Seeded fuzz aimed at the last-level table: forests of one to four classes, two to twelve levels, sides that turn into
    leaves with probability 0 ... 0.3 per level, few levels pinned in LDS (so level D-1 is walked from the table), now and then
    a deepest node the table cannot serve; knob forced on / left alone, random launch geometry, reduce, scale, filter,
    pre-fill -- bit-exact against the C oracle."""
    rounds = int(os.environ.get("RDF_LAST_LEVEL_FUZZ_ROUNDS", "50"))
    rng = np.random.default_rng(int(os.environ.get("RDF_FUZZ_SEED", "20211003")) + 29)
    lib = gpu_runtime.lib
    synth = rdf.synth
    try:
        for it in range(rounds):
            T, D, C = int(rng.integers(1, 10)), int(rng.integers(2, 13)), int(rng.integers(1, 5))
            p = float(rng.choice([0.0, 0.0, 0.02, 0.1, 0.3]))
            forest = np.stack([synth.full_tree(5000 + 10 * it + k, D, C) if p == 0 else
                               synth.trained_like_tree(5000 + 10 * it + k, D, C, leaf_prob=p, min_leaf_level=int(rng.integers(0, 4)))
                               for k in range(T)])
            first = (1 << (D - 1)) - 1
            forest[:, first:, 7:] = rng.random((T, first + 1, 2 * C), dtype=np.float32)
            spoil = rng.random()
            if spoil < 0.1:
                forest[int(rng.integers(0, T)), first + int(rng.integers(0, first + 1)), 5 + int(rng.integers(0, 2))] = -1.0
            elif spoil < 0.2:
                forest[int(rng.integers(0, T)), first + int(rng.integers(0, first + 1)), int(rng.integers(0, 4))] = float(rng.choice([3e7, np.nan, 1e-40]))
            big = rng.random() < 0.15
            n, h, w = (int(rng.integers(30, 50)), 240, 424) if big else (int(rng.integers(1, 4)), int(rng.integers(1, 90)), int(rng.integers(1, 200)))
            r = int(rng.choice([1, 1, 2, 3]))
            s = float(rng.choice([1.0, 0.5, 1.5]))
            kinds = [str(k) for k in rng.choice(["dense", "live"], size=n)]
            depth = synth.frames(kinds, 9000 + it, h, w)
            depth[rng.random(depth.shape) < 0.02] = 0
            use_filter = rng.random() < 0.3
            filt = rng.integers(0, 3, size=(n, h // r, w // r)).astype(np.uint16) if use_filter else None
            prefill = int(rng.choice([65535, 0]))
            lib.rdf_set_last_level_table(int(rng.choice([1, 1, -1])))
            lib.rdf_set_lds_levels(int(rng.integers(0, D)))
            lib.rdf_set_block_threads(int(rng.choice([0, 256, 512])))
            lib.rdf_set_halo(int(rng.choice([-1, 0, 16, 40])))
            lib.rdf_set_group(int(rng.choice([0, 0, 1, 2, 3, 4])))
            lib.rdf_set_rows_per_wave(int(rng.choice([0, 1, 2, 4])))
            lib.rdf_set_tree_waves(int(rng.choice([-1, 0])))
            want = np.full((n, h // r, w // r), prefill, np.uint16)
            oracle.eval_forest(depth, forest, want, r, filt, 2 if use_filter else None, s)
            got = _gpu_forest(rdf, evs["packed"], depth, forest, prefill, r, filt, 2 if use_filter else None, s)
            assert np.array_equal(got, want), f"iteration {it}: T{T} D{D} C{C} p{p} {n}x{h}x{w} r{r} s{s}: {(got != want).sum()} pixels"
    finally:
        lib.rdf_set_last_level_table(-1)
        lib.rdf_set_lds_levels(-1)
        lib.rdf_set_block_threads(0)
        lib.rdf_set_halo(-1)
        lib.rdf_set_group(0)
        lib.rdf_set_rows_per_wave(0)
        lib.rdf_set_tree_waves(-1)


@pytest.mark.parametrize("path", ["packed", "direct"])
def test_filled_entry_point_equals_fill_then_evaluate(path, rdf, evs, oracle, gpu_runtime):
    """rdf_eval_forest_packed_filled / get_labels_forest_filled: the caller's 65535 fill folded into the evaluation, on big and
    small launches, with and without a filter, at labels_reduce 1 and 2, from a label buffer full of something else."""
    synth = rdf.synth
    f_np = synth.forest(4, 9, 4, "trained", 310)
    forest = rdf.DecisionForest.from_numpy(f_np)
    rng = np.random.default_rng(3)
    for n, h, w in ((40, 240, 424), (2, 70, 90)):
        frames = synth.frames(["dense", "live"] * (n // 2), 77 + n, h, w)
        frames[rng.random(frames.shape) < 0.05] = 0
        depth = rdf.to_device(frames)
        for r in (1, 2):
            for use_filter in (False, True):
                filt = rng.integers(0, 3, size=(n, h // r, w // r)).astype(np.uint16) if use_filter else None
                want = np.full((n, h // r, w // r), 65535, np.uint16)
                oracle.eval_forest(frames, f_np, want, r, filt, 2 if use_filter else None)
                labels = rdf.DeviceArray(want.shape, np.uint16).fill(4242)
                evs[path].get_labels_forest_filled(forest, depth, labels, r, rdf.to_device(filt) if use_filter else None,
                                                   2 if use_filter else None)
                assert np.array_equal(labels.get(), want), (n, r, use_filter, int((labels.get() != want).sum()))
    # a degenerate forest (depth 0): nothing to evaluate, every pixel 65535
    lib = gpu_runtime.lib
    labels = rdf.DeviceArray((2, 70, 90), np.uint16).fill(7)
    assert lib.rdf_eval_forest_packed_filled(rdf.to_device(frames).ptr, 2, 90, 70, None, None, 0, 0, 4, None, -1, labels.ptr, 1,
                                             gpu_runtime.stream()) == 0
    assert np.all(labels.get() == 65535)


# ---- deep blocks (k_eval_forest<..., DEEP>): the levels no cache holds, three to a 128-byte line ----

def _deep_forest(rdf, topology, T, D, C, first_tree=0):
    if topology == "balanced":
        return rdf.synth.forest(T, D, C, "balanced", first_tree, calib=rdf.synth.calibration_frames(4, 120, 212))
    return rdf.synth.forest(T, D, C, topology, first_tree)


@pytest.mark.parametrize("topology,T,D,C", [("balanced", 4, 13, 4), ("trained", 4, 14, 4), ("balanced", 3, 12, 7), ("full", 8, 11, 3),
                                            ("balanced", 1, 12, 2), ("trained", 2, 13, 8), ("trained", 6, 12, 5), ("balanced", 5, 11, 4)])
def test_deep_blocks_give_the_oracles_labels(rdf, evs, oracle, gpu_runtime, topology, T, D, C):
    """Every block root level as the place where the deep blocks take over, both workgroup sizes: the labels are the
    oracle's (and the heap-order path's)."""
    lib = gpu_runtime.lib
    forest = _deep_forest(rdf, topology, T, D, C, first_tree=40)
    depth = rdf.synth.frames(["dense", "live", "dense"], 700, 120, 212)
    want = np.full(depth.shape, 65535, np.uint16)
    oracle.eval_forest(depth, forest, want)
    last = 2 if C <= 4 else 1
    try:
        for block in (0, 256, 512):
            lib.rdf_set_block_threads(block)
            for level in [0] + list(range((D - last) % 3, D - last + 1, 3)) + [D + 3]:
                lib.rdf_set_deep_from(level)
                got = _gpu_forest(rdf, evs["packed"], depth, forest, 65535)
                assert np.array_equal(got, want), (block, level, int((got != want).sum()))
    finally:
        lib.rdf_set_block_threads(0)
        lib.rdf_set_deep_from(-1)


def test_deep_blocks_under_every_lds_budget(rdf, evs, oracle, gpu_runtime):
    """The LDS budget knob is for the depth tile and the node table; a launch that walks deep blocks adds 8 KB of slab per wave
    on top.  Whatever the knob says (a whole CU's 160 KB, next to nothing), the workgroup fits a CU and the labels are the
    oracle's -- round 5's first version let budget + slabs exceed 160 KB and the launch failed."""
    lib = gpu_runtime.lib
    forest = _deep_forest(rdf, "balanced", 4, 12, 4, first_tree=61)
    depth = rdf.synth.frames(["dense", "live"], 1700, 120, 212)
    want = np.full(depth.shape, 65535, np.uint16)
    oracle.eval_forest(depth, forest, want)
    try:
        lib.rdf_set_deep_from(7)
        for block in (256, 512):
            for budget, halo in ((163840, 96), (163840, -1), (81920, 40), (4096, -1), (1, 0)):
                lib.rdf_set_block_threads(block)
                lib.rdf_set_lds_budget_bytes(budget)
                lib.rdf_set_halo(halo)
                got = _gpu_forest(rdf, evs["packed"], depth, forest, 65535)
                assert np.array_equal(got, want), (block, budget, halo, int((got != want).sum()))
    finally:
        lib.rdf_set_deep_from(-1)
        lib.rdf_set_block_threads(0)
        lib.rdf_set_lds_budget_bytes(0)
        lib.rdf_set_halo(-1)


@pytest.mark.parametrize("r,use_filter,prefill", [(1, True, 65535), (2, False, 0), (2, True, 7), (3, False, 65535)])
def test_deep_blocks_with_filter_reduce_and_prefill(rdf, evs, oracle, gpu_runtime, r, use_filter, prefill):
    lib = gpu_runtime.lib
    rng = np.random.default_rng(5)
    forest = _deep_forest(rdf, "balanced", 4, 12, 4, first_tree=3)
    depth = rdf.synth.frames(["live", "dense"], 900, 96, 200)
    filt = rng.integers(0, 3, size=(2, 96 // r, 200 // r)).astype(np.uint16) if use_filter else None
    want = np.full((2, 96 // r, 200 // r), prefill, np.uint16)
    oracle.eval_forest(depth, forest, want, r, filt, 1 if use_filter else None, 0.5)
    try:
        for level in (4, 7, 10):
            lib.rdf_set_deep_from(level)
            got = _gpu_forest(rdf, evs["packed"], depth, forest, prefill, r, filt, 1 if use_filter else None, 0.5)
            assert np.array_equal(got, want), (level, int((got != want).sum()))
    finally:
        lib.rdf_set_deep_from(-1)


def test_deep_blocks_step_aside_for_nodes_they_cannot_serve(rdf, evs, oracle, gpu_runtime):
    """A node that needs the IEEE divide below the take-over level, or a node of the last level that is not a plain
    two-leaf node, makes the kernel take the heap-order path (the table's trailer says so): same labels."""
    lib = gpu_runtime.lib
    depth = rdf.synth.frames(["dense", "live"], 1300, 100, 180)
    base = _deep_forest(rdf, "balanced", 4, 12, 4, first_tree=9)
    cases = []
    f1 = base.copy()
    f1[1, (1 << 9) - 1 + 5, 0] = 3.0e7          # level 9: numerator beyond the integer record
    cases.append(("exact node on level 9", f1))
    f2 = base.copy()
    f2[2, (1 << 11) - 1 + 17, 5] = -1.0         # level 11 (the last): left side says "continue" (tree_eval.cu:95-128: no leaf)
    cases.append(("continue flag on the last level", f2))
    f3 = base.copy()
    f3[0, 0, 1] = np.inf                          # root: blocks from a deeper level stay usable
    cases.append(("exact node on level 0", f3))
    try:
        for name, forest in cases:
            want = np.full(depth.shape, 65535, np.uint16)
            oracle.eval_forest(depth, forest, want)
            for level in (0, 4, 7, 10):
                lib.rdf_set_deep_from(level)
                got = _gpu_forest(rdf, evs["packed"], depth, forest, 65535)
                assert np.array_equal(got, want), (name, level, int((got != want).sum()))
        lib.rdf_set_force_exact(1)
        lib.rdf_set_deep_from(7)
        want = np.full(depth.shape, 65535, np.uint16)
        oracle.eval_forest(depth, base, want)
        assert np.array_equal(_gpu_forest(rdf, evs["packed"], depth, base, 65535), want)
    finally:
        lib.rdf_set_force_exact(0)
        lib.rdf_set_deep_from(-1)


def test_forest_tune_picks_a_candidate_and_keeps_the_labels(rdf, evs, oracle, gpu_runtime):
    """rdf_forest_tune times every candidate on the caller's frames, remembers the fastest for that packed table, and a
    re-pack forgets it; labels never depend on the choice."""
    lib = gpu_runtime.lib
    forest_np = _deep_forest(rdf, "balanced", 4, 14, 4, first_tree=21)
    depth_np = rdf.synth.frames(["dense", "live", "dense", "dense"], 1500, 120, 212)
    f = rdf.DecisionForest.from_numpy(forest_np)
    depth = rdf.to_device(depth_np)
    res = f.tune(depth)
    assert set(res["tried"]) == {0, 6, 9, 12} and res["deep_from"] in res["tried"], res
    assert all(ms > 0 for ms in res["tried"].values()), res
    want = np.full(depth_np.shape, 65535, np.uint16)
    oracle.eval_forest(depth_np, forest_np, want)
    out = rdf.DeviceArray(depth_np.shape, np.uint16).fill(65535)
    evs["packed"].get_labels_forest(f, depth, out)
    assert np.array_equal(out.get(), want)
    # an explicit choice for this table, then "forget"
    packed = f.packed(1.0)
    for level in (9, 0, -1):
        assert lib.rdf_forest_set_deep_from(packed.ptr, level) == 0
        out.fill(65535)
        evs["packed"].get_labels_forest(f, depth, out)
        assert np.array_equal(out.get(), want), level
    assert lib.rdf_forest_set_deep_from(None, 3) != 0


def test_balanced_topology_config2_full_size(rdf, evs, oracle, gpu_runtime):
    """Config 2's frames on a T4/D20 forest whose deep levels are occupied (synth's balanced topology: median
    thresholds over calibration frames), heap-order path and deep blocks: bit-exact, and the batch really visits most of the
    deepest level."""
    lib = gpu_runtime.lib
    forest = rdf.synth.forest(4, 20, 4, "balanced")
    depth = rdf.synth.frames(["dense", "live"], 0, 480, 848)
    want = np.full(depth.shape, 65535, np.uint16)
    st = np.zeros(3, np.uint64)
    oracle.eval_forest(depth, forest, want, stats=st)
    assert int(st[1]) == int(st[0]) * 4 * 20 and int(st[2]) == int(st[0]) * 4      # every walk reaches level D-1
    distinct = oracle.distinct_nodes_per_level(depth, forest)
    assert distinct[:, 19].min() > 0.45 * (1 << 19), distinct[:, 19]              # (the "full" topology: 1.7 %)
    f = rdf.DecisionForest.from_numpy(forest)
    d = rdf.to_device(depth)
    try:
        for level in (0, 12, 15, 18, -1):
            lib.rdf_set_deep_from(level)
            out = rdf.DeviceArray(depth.shape, np.uint16).fill(65535)
            evs["packed"].get_labels_forest(f, d, out)
            got = out.get()
            assert np.array_equal(got, want), (level, int((got != want).sum()))
    finally:
        lib.rdf_set_deep_from(-1)


def test_packed_stats_count_what_the_timed_launch_does(rdf, evs, oracle, gpu_runtime):
    """rdf_eval_forest_packed_stats: the packed launch with counters on.  Pixels, node visits and leaves are the oracle's; the
    line counters follow the table the launch walks (heap-order records / last-level records / deep blocks)."""
    lib = gpu_runtime.lib
    forest_np = _deep_forest(rdf, "balanced", 4, 14, 4, first_tree=77)
    depth_np = rdf.synth.frames(["dense", "live", "dense"], 2100, 240, 424)
    want = np.full(depth_np.shape, 65535, np.uint16)
    st = np.zeros(3, np.uint64)
    oracle.eval_forest(depth_np, forest_np, want, stats=st)
    f = rdf.DecisionForest.from_numpy(forest_np)
    depth = rdf.to_device(depth_np)
    packed = f.packed(1.0)
    seen = {}
    try:
        for level in (0, 9, 12):
            lib.rdf_set_deep_from(level)
            out = rdf.DeviceArray(depth_np.shape, np.uint16).fill(65535)
            st8 = rdf.DeviceArray((8,), np.uint64).fill(0)
            rc = lib.rdf_eval_forest_packed_stats(depth.ptr, 3, 424, 240, packed.ptr, f.forest_cu.ptr, 4, 14, 4, out.ptr, 1, st8.ptr,
                                                  gpu_runtime.stream())
            assert rc == 0
            v = [int(x) for x in st8.get()]
            assert np.array_equal(out.get(), want)
            assert v[0:3] == [int(st[0]), int(st[1]), int(st[2])], (level, v)
            px, visits = v[0], v[1]
            assert 0 < v[3] < visits and v[3] % 1 == 0                 # the top levels come from LDS
            assert v[6] > 0                                           # some probes leave the staged tile
            if level == 0:
                assert v[7] == 0 and 0 < v[4] <= visits - v[3]        # one line or less per record read from global memory
            else:
                # blocks: one per walking (pixel, tree) and block level from `level` on; every walk reaches the last level
                n_blocks = len(range(level, 12, 3)) + 1
                assert v[7] == px * 4 * n_blocks, (level, v[7], px * 4 * n_blocks)
                assert v[5] == 0                                      # the leaves come with the last block
            seen[level] = v
        assert seen[9][4] < seen[12][4] < seen[0][4]                  # fewer heap-order records the earlier the blocks take over
        # refused: more than four classes, a NULL table
        assert lib.rdf_eval_forest_packed_stats(depth.ptr, 3, 424, 240, None, f.forest_cu.ptr, 4, 14, 4, out.ptr, 1, st8.ptr, 0) != 0
    finally:
        lib.rdf_set_deep_from(-1)


def test_exact_numerators_come_from_the_callers_forest(rdf, evs, oracle, gpu_runtime):
    """A packed table keeps no copy of the fp32 numerators: a node whose numerators the integer record cannot hold (huge,
    denormal, non-finite) reads them from the reference-layout forest the caller passes along, scaled as rdf_forest_pack
    scaled them.  Such a table refuses an evaluation without the forest; a table without such nodes does not need it."""
    lib = gpu_runtime.lib
    depth_np = rdf.synth.frames(["dense", "live"], 3100, 96, 160)
    depth = rdf.to_device(depth_np)
    plain = rdf.synth.forest(3, 9, 4, "full", 5)
    wild = plain.copy()
    wild[1, 7, 0] = 3.0e7
    wild[2, 100, 3] = 1e-41
    wild[0, 300, 2] = np.inf
    for forest_np, needs in ((plain, False), (wild, True)):
        for s_ in (1.0, 0.37):
            f = rdf.DecisionForest.from_numpy(forest_np)
            packed = f.packed(s_)
            want = np.full(depth_np.shape, 65535, np.uint16)
            oracle.eval_forest(depth_np, forest_np, want, 1, None, None, s_)
            out = rdf.DeviceArray(depth_np.shape, np.uint16).fill(65535)
            rc = lib.rdf_eval_forest_packed(depth.ptr, 2, 160, 96, packed.ptr, f.forest_cu.ptr, 3, 9, 4, None, -1, out.ptr, 1, gpu_runtime.stream())
            assert rc == 0 and np.array_equal(out.get(), want), (needs, s_)
            out.fill(65535)
            rc = lib.rdf_eval_forest_packed(depth.ptr, 2, 160, 96, packed.ptr, None, 3, 9, 4, None, -1, out.ptr, 1, gpu_runtime.stream())
            if needs:
                assert rc == -2                          # RDF_ERR_NULL_PTR
            else:
                assert rc == 0 and np.array_equal(out.get(), want)


def _blocks_walked(rdf, gpu_runtime, f, depth, shape):
    """Deep blocks the launch rdf_eval_forest_packed would make for this batch loads (rdf_eval_forest_packed_stats, counter 7)."""
    lib = gpu_runtime.lib
    n, h, w = shape
    st8 = rdf.DeviceArray((8,), np.uint64).fill(0)
    tmp = rdf.DeviceArray(shape, np.uint16).fill(65535)
    rc = lib.rdf_eval_forest_packed_stats(depth.ptr, n, w, h, f.packed(1.0).ptr, f.forest_cu.ptr, int(f.num_trees), int(f.max_depth),
                                          int(f.num_classes), tmp.ptr, 1, st8.ptr, gpu_runtime.stream())
    assert rc == 0
    return int(st8.get()[7])


def test_first_big_evaluation_tunes_a_big_forest_once(rdf, oracle, gpu_runtime, caplog, monkeypatch):
    """DecisionTreeEvaluator's safety net: a big packed forest whose table carries no choice is tuned at its first batch-sized
    evaluation (and only then), with ONE log line; small forests, small launches, evaluators with auto_tune off and processes
    with RDF_AUTO_TUNE=0 are left alone; a failing tune does not fail the evaluation."""
    import logging
    forest_np = rdf.synth.forest(4, 19, 4, "full", 31)              # 32 MB of hot records: the size from which the choice is open
    depth_np = rdf.synth.mixed_batch(10, 5200, 480, 848)
    depth = rdf.to_device(depth_np)
    f = rdf.DecisionForest.from_numpy(forest_np)
    ev = rdf.DecisionTreeEvaluator()
    out = rdf.DeviceArray(depth_np.shape, np.uint16).fill(65535)
    ev.get_labels_forest(f, depth[0:2], out[0:2])                   # two frames: not a batch
    assert not f.__dict__.get("_tuned") and f.deep_from() is None
    with caplog.at_level(logging.WARNING, logger="rdf_hip"):
        ev.get_labels_forest(f, depth, out)
    lines = [r.getMessage() for r in caplog.records if r.name == "rdf_hip"]
    assert len(lines) == 1 and "auto-tuned" in lines[0] and "T4/D19/C4" in lines[0] and "RDF_AUTO_TUNE=0" in lines[0], lines
    tuned = f.__dict__["_tuned"][1.0]
    assert tuned["deep_from"] in tuned["tried"] and 0 in tuned["tried"] and len(tuned["tried"]) >= 4
    assert f.deep_from() == tuned["deep_from"]                      # the choice is in the table
    want = np.full((3,) + depth_np.shape[1:], 65535, np.uint16)
    oracle.eval_forest(depth_np[0:3], forest_np, want)
    assert np.array_equal(out[0:3].get(), want)
    before = dict(f.__dict__["_tuned"])
    caplog.clear()
    with caplog.at_level(logging.WARNING, logger="rdf_hip"):
        ev.get_labels_forest(f, depth, out)
        rdf.DecisionTreeEvaluator().get_labels_forest(f, depth, out)    # (another evaluator: the forest, not the evaluator, remembers)
    assert f.__dict__["_tuned"] == before and not caplog.records    # once
    small = rdf.DecisionForest.from_numpy(rdf.synth.forest(4, 12, 4, "full", 32))
    ev.get_labels_forest(small, depth, out)
    assert not small.__dict__.get("_tuned")
    f2 = rdf.DecisionForest.from_numpy(forest_np)
    ev_off = rdf.DecisionTreeEvaluator()
    ev_off.auto_tune = False
    ev_off.get_labels_forest(f2, depth, out)
    assert not f2.__dict__.get("_tuned") and f2.deep_from() is None
    monkeypatch.setenv("RDF_AUTO_TUNE", "0")
    ev.get_labels_forest(f2, depth, out)
    assert not f2.__dict__.get("_tuned") and f2.deep_from() is None
    monkeypatch.delenv("RDF_AUTO_TUNE")
    # a tune that fails (here: made to raise) is logged and the evaluation goes on from the heap-order records
    def broken(*a_, **k_):
        raise MemoryError("no room for the scratch label map")
    f2.tune = broken
    out.fill(65535)
    caplog.clear()
    with caplog.at_level(logging.WARNING, logger="rdf_hip"):
        ev.get_labels_forest(f2, depth, out)
    assert "MemoryError" in f2.__dict__["_tuned"][1.0]["error"] and len(caplog.records) == 1 and "failed" in caplog.records[0].getMessage()
    assert np.array_equal(out[0:3].get(), want)


def test_untuned_table_walks_heap_order_records_and_a_choice_travels_with_the_table(rdf, oracle, gpu_runtime):
    """Nobody chose: a big forest's batch launch loads no deep block (version 3 of the library switched by size alone).  A
    choice made for a table is written INTO the table: the table's bytes adopted by another DecisionForest carry it, the
    adopting side neither packs nor tunes, and the labels are the oracle's."""
    import logging
    lib = gpu_runtime.lib
    forest_np = rdf.synth.forest(4, 19, 4, "full", 57)              # 32 MB of hot records
    depth_np = rdf.synth.mixed_batch(10, 6400, 480, 848)
    depth = rdf.to_device(depth_np)
    want = np.full((2,) + depth_np.shape[1:], 65535, np.uint16)
    oracle.eval_forest(depth_np[0:2], forest_np, want)
    f = rdf.DecisionForest.from_numpy(forest_np)
    assert f.packed(1.0) is not None and f.deep_from() is None
    assert _blocks_walked(rdf, gpu_runtime, f, depth, depth_np.shape) == 0
    assert lib.rdf_forest_set_deep_from(f.packed(1.0).ptr, 14) == 0
    assert f.deep_from() == 14 and _blocks_walked(rdf, gpu_runtime, f, depth, depth_np.shape) > 0
    table = f.packed_bytes(1.0)
    # ... the adopting side: same forest, the table as bytes (a file, another process)
    g = rdf.DecisionForest.from_numpy(forest_np)
    g.adopt_packed(table, 1.0)
    ev = rdf.DecisionTreeEvaluator()
    out = rdf.DeviceArray(depth_np.shape, np.uint16).fill(65535)
    ev.get_labels_forest(g, depth, out)                             # batch-sized, auto-tune on: the table's own choice stands
    assert g.__dict__["_tuned"][1.0] == {"deep_from": 14, "tried": None, "source": "the packed table"}
    assert g.deep_from() == 14 and _blocks_walked(rdf, gpu_runtime, g, depth, depth_np.shape) > 0
    assert np.array_equal(out[0:2].get(), want)
    # "no choice" written through as well
    assert lib.rdf_forest_set_deep_from(g.packed(1.0).ptr, -1) == 0 and g.deep_from() is None
    assert _blocks_walked(rdf, gpu_runtime, g, depth, depth_np.shape) == 0
    assert rdf.DecisionForest.from_numpy(forest_np).adopt_packed(g.packed_bytes(1.0), 1.0) is not None
    # a choice made for a table the process has not looked at yet (no info block read so far) goes into the table at the first look
    k = rdf.DecisionForest.from_numpy(forest_np)
    k.adopt_packed(g.packed_bytes(1.0), 1.0)                        # (carries "no choice")
    assert lib.rdf_forest_set_deep_from(k.packed(1.0).ptr, 11) == 0
    assert k.deep_from() == 11
    again = rdf.DecisionForest.from_numpy(forest_np)
    again.adopt_packed(k.packed_bytes(1.0), 1.0)
    assert again.deep_from() == 11
    # memory that is not a packed table of this shape is refused at the first look -- which adopt_packed takes at once (round 6):
    # zeros, a table of another shape's size cut to fit, a table packed for another scale
    junk = rdf.DecisionForest.from_numpy(forest_np)
    with pytest.raises(ValueError, match="not a packed table"):
        junk.adopt_packed(np.zeros_like(table), 1.0)
    assert junk._packed == {} and junk.deep_from() is None
    with pytest.raises(ValueError, match="bytes"):
        junk.adopt_packed(table[:-128], 1.0)
    with pytest.raises(ValueError, match="packed for scale_factor 1.0"):
        junk.adopt_packed(table, 0.5)
    assert junk._packed == {}
    ev.get_labels_forest(junk, depth[0:1], out[0:1])                # (packs its own table and goes on)
    assert np.array_equal(out[0:1].get(), want[0:1])


def test_a_table_that_lands_on_a_known_address_is_read_afresh(rdf, oracle, gpu_runtime):
    """The library remembers a table's scale and exact-node count per (device, address).  A DIFFERENT table written to a known
    address (adopt_packed: upload, no rdf_forest_pack) is evaluated with ITS scale: adopt_packed forgets the address first
    (rdf_forest_forget) -- without that the old table's scale would be used silently."""
    lib = gpu_runtime.lib
    forest_np = rdf.synth.forest(3, 10, 4, "trained", 5)
    depth_np = rdf.synth.frames(["dense", "live"], 7100, 96, 160)
    depth = rdf.to_device(depth_np)
    f = rdf.DecisionForest.from_numpy(forest_np)
    half = rdf.DecisionForest.from_numpy(forest_np)
    table_half = half.packed_bytes(0.5)                            # the same forest packed for scale 0.5
    buf = f.packed(1.0)
    out = rdf.DeviceArray(depth_np.shape, np.uint16).fill(65535)
    rc = lib.rdf_eval_forest_packed(depth.ptr, 2, 160, 96, buf.ptr, f.forest_cu.ptr, 3, 10, 4, None, -1, out.ptr, 1, gpu_runtime.stream())
    assert rc == 0
    want1 = np.full(depth_np.shape, 65535, np.uint16)
    oracle.eval_forest(depth_np, forest_np, want1, 1, None, None, 1.0)
    assert np.array_equal(out.get(), want1)
    with pytest.raises(ValueError, match="packed for scale_factor 0.5"):      # (round 6: the key must be the table's own scale)
        f.adopt_packed(table_half, 1.0)
    f._packed[1.0] = ((id(f.forest_cu), f.forest_cu.version), buf)  # the same memory once more: adopt into the known address
    f._packed[0.5] = f._packed.pop(1.0)
    same_buf = f.adopt_packed(table_half, 0.5)
    assert same_buf.ptr == buf.ptr
    scale = __import__("ctypes").c_float(0)
    assert lib.rdf_forest_info(buf.ptr, 3, 10, 4, gpu_runtime.stream(), None, None, __import__("ctypes").byref(scale)) == 0 and scale.value == 0.5
    # a node with a huge numerator takes the exact path, which multiplies by the table's scale: make one so the scale matters
    wild = forest_np.copy()
    wild[0, 0, 0] = 3.0e7
    fw = rdf.DecisionForest.from_numpy(wild)
    fw.adopt_packed(rdf.DecisionForest.from_numpy(wild).packed_bytes(0.5), 0.5)
    out.fill(65535)
    rc = lib.rdf_eval_forest_packed(depth.ptr, 2, 160, 96, fw.packed(0.5).ptr, fw.forest_cu.ptr, 3, 10, 4, None, -1, out.ptr, 1, gpu_runtime.stream())
    want_half = np.full(depth_np.shape, 65535, np.uint16)
    oracle.eval_forest(depth_np, wild, want_half, 1, None, None, 0.5)
    assert rc == 0 and np.array_equal(out.get(), want_half)


def test_a_table_copied_over_a_known_address_without_forget_is_reported_stale(rdf, oracle, gpu_runtime):
    """What a C consumer can do wrong (round 6): write ANOTHER packed table over an address the library has evaluated from --
    a plain copy, no rdf_forest_pack, no rdf_forest_forget.  Every packing carries a generation number in its info block and
    every launch the number the host remembers; the kernel compares them and raises the device's stale flag (pinned host
    memory: no synchronous read), the NEXT call returns RDF_ERR_STALE and the library has forgotten what it knew, the call
    after that reads the info block afresh.  The launch that met the foreign table already used the TABLE's scale (the kernel
    reads it from the info block, not from the host's memory), so its labels are the new table's."""
    import ctypes
    lib, st = gpu_runtime.lib, gpu_runtime.stream()
    forest_np = rdf.synth.forest(3, 10, 4, "trained", 5)
    wild = forest_np.copy()
    wild[0, 0, 0] = 3.0e7                                           # an exact node at the root of tree 0: the scale matters
    depth_np = rdf.synth.frames(["dense", "live"], 7100, 96, 160)
    depth = rdf.to_device(depth_np)
    out = rdf.DeviceArray(depth_np.shape, np.uint16).fill(65535)
    want = {}
    for s in (1.0, 0.5):
        want[s] = np.full(depth_np.shape, 65535, np.uint16)
        oracle.eval_forest(depth_np, wild, want[s], 1, None, None, s)
    assert not np.array_equal(want[1.0], want[0.5])
    f = rdf.DecisionForest.from_numpy(wild)
    buf = f.packed(1.0)

    def call(forest_ptr=f.forest_cu.ptr):
        out.fill(65535)
        rc = lib.rdf_eval_forest_packed(depth.ptr, 2, 160, 96, buf.ptr, forest_ptr, 3, 10, 4, None, -1, out.ptr, 1, st)
        return rc, out.get()                                        # (.get() synchronises: the kernel has run)

    rc, got = call()
    assert rc == 0 and np.array_equal(got, want[1.0])
    other = rdf.DecisionForest.from_numpy(wild).packed_bytes(0.5)   # the same forest packed for another scale, elsewhere
    buf.set(other)                                                  # a raw copy over the known address; the library is not told
    rc, got = call()
    assert rc == 0 and np.array_equal(got, want[0.5])               # the table's own scale
    rc, _ = call()
    assert rc == -7 and b"rdf_forest_forget" in lib.rdf_error_string(-7)
    rc, got = call()
    assert rc == 0 and np.array_equal(got, want[0.5])
    scale = ctypes.c_float(0)
    assert lib.rdf_forest_info(buf.ptr, 3, 10, 4, st, None, None, ctypes.byref(scale)) == 0 and scale.value == 0.5
    # the other thing the host remembers is whether a table needs the caller's forest: a tame table evaluated without one ...
    tame = rdf.DecisionForest.from_numpy(forest_np)
    buf.set(tame.packed_bytes(1.0))
    assert lib.rdf_forest_forget(buf.ptr) == 0                      # (told this time)
    rc, got = call(None)
    tame_want = np.full(depth_np.shape, 65535, np.uint16)
    oracle.eval_forest(depth_np, forest_np, tame_want)
    assert rc == 0 and np.array_equal(got, tame_want)
    # ... and the wild one copied over it without a word: no fault (the kernel has no forest to read), the flag, then the refusal
    buf.set(other)
    rc, _ = call(None)
    assert rc == 0
    rc, _ = call(None)
    assert rc == -7
    rc, _ = call(None)
    assert rc == -2                                                 # RDF_ERR_NULL_PTR: this table needs the forest
    rc, got = call()
    assert rc == 0 and np.array_equal(got, want[0.5])
    f._forget(buf)


def test_split_launch_shares_one_tile_queue(rdf, oracle, gpu_runtime):
    """rdf_eval_forest_packed_split (round 6): a main launch on a CU-masked stream and a helper launch on an ordinary stream
    pull from ONE tile queue.  Whenever the helper arrives -- at once, a few hundred microseconds into the main launch, after
    the main launch has finished -- every tile is evaluated exactly once: the oracle's labels, with and without the fused
    fill, on back-to-back steps that alternate the queue tag, and for a launch too small to split."""
    import ctypes
    import torch
    lib = gpu_runtime.lib
    forest_np = rdf.synth.forest(4, 12, 4, "trained", 31)
    depth_np = rdf.synth.mixed_batch(24, 8800, 480, 848)
    want = np.full(depth_np.shape, 65535, np.uint16)
    oracle.eval_forest(depth_np, forest_np, want)
    f = rdf.DecisionForest.from_numpy(forest_np)
    f.packed(1.0)
    depth = rdf.to_device(depth_np)
    ev = rdf.DecisionTreeEvaluator()
    ev.auto_tune = False
    h = ctypes.c_void_p()
    assert lib.rdf_stream_create_with_reserved_cus(ctypes.byref(h), 32) == 0
    main = torch.cuda.ExternalStream(h.value)
    helper = torch.cuda.Stream()
    t_start = rdf.DeviceArray((64,), np.uint64).fill(0)
    outs = [rdf.DeviceArray(depth_np.shape, np.uint16) for _ in range(3)]
    try:
        for delay_ticks, fill in ((0, False), (30_000, False), (30_000, True), (3_000_000, False)):     # 0, 0.3 ms, 30 ms
            out = outs[0].fill(0 if fill else 65535)
            torch.cuda.synchronize()
            if delay_ticks:         # something that keeps the helper stream (and the 32 reserved CUs) busy for a while
                assert lib.rdf_debug_fat_kernel(16, delay_ticks, t_start.ptr, ctypes.c_void_p(helper.cuda_stream)) == 0
            with torch.cuda.stream(main):
                n = ev.get_labels_forest_split(f, depth, out, helper.cuda_stream, 32, queue_tag=0, fill_untouched=fill)
            torch.cuda.synchronize()
            got = out.get()
            assert n > 0 and np.array_equal(got, want), (delay_ticks, fill, n, int((got != want).sum()))
        # three steps back to back, tags alternating, the next step's main launch ordered after the helper of the step before
        for o in outs:
            o.fill(65535)
        torch.cuda.synchronize()
        for s, o in enumerate(outs):
            if s == 1:
                assert lib.rdf_debug_fat_kernel(16, 50_000, t_start.ptr, ctypes.c_void_p(helper.cuda_stream)) == 0
            with torch.cuda.stream(main):
                helper.wait_stream(main)
                ev.get_labels_forest_split(f, depth, o, helper.cuda_stream, 32, queue_tag=s & 1)
                main.wait_stream(helper)
        torch.cuda.synchronize()
        for s, o in enumerate(outs):
            got = o.get()
            assert np.array_equal(got, want), (s, int((got != want).sum()))
        # the same stream's ordinary launches afterwards: the split slots are their own
        out = outs[0].fill(65535)
        with torch.cuda.stream(main):
            ev.get_labels_forest(f, depth, out)
        torch.cuda.synchronize()
        assert np.array_equal(out.get(), want)
        # one frame (a small launch: 256-thread workgroups, narrow tiles; 1 590 tiles on the masked stream's 1 120 slots, so it is
        # split too).  The helper stream is the CALLER's to order: it must come after whatever wrote the inputs and the pre-fill
        # (here: the fill, on the current stream) -- a helper that ran ahead of the fill would see its labels overwritten
        one = rdf.DeviceArray((1,) + depth_np.shape[1:], np.uint16).fill(65535)
        with torch.cuda.stream(main):
            helper.wait_stream(main)
            n = ev.get_labels_forest_split(f, depth[3:4], one, helper.cuda_stream, 32)
        torch.cuda.synchronize()
        assert n > 0 and np.array_equal(one.get(), want[3:4])
        # a filtered launch through the C entry point (pixel lists; at labels_reduce 2 such a launch takes 512-thread workgroups)
        lh, lw = 480 // 2, 848 // 2
        filt_np = (np.arange(24 * lh * lw).reshape(24, lh, lw) % 3).astype(np.uint16)
        want_f = np.full((24, lh, lw), 65535, np.uint16)
        oracle.eval_forest(depth_np, forest_np, want_f, 2, filt_np, 1)
        filt = rdf.to_device(filt_np)
        out_f = rdf.DeviceArray((24, lh, lw), np.uint16).fill(0)
        n_h = ctypes.c_int(0)
        with torch.cuda.stream(main):
            helper.wait_stream(main)            # (the fill above)
            rc = lib.rdf_eval_forest_packed_split(depth.ptr, 24, 848, 480, f.packed(1.0).ptr, f.forest_cu.ptr, 4, 12, 4, filt.ptr, 1, out_f.ptr, 2, 1,
                                                  ctypes.c_void_p(main.cuda_stream), ctypes.c_void_p(helper.cuda_stream), 32, 1, ctypes.byref(n_h))
        torch.cuda.synchronize()
        got = out_f.get()
        assert rc == 0 and np.array_equal(got, want_f), (rc, n_h.value, int((got != want_f).sum()))
        # without the dynamic tile queue (static striding, one workgroup per tile) there is nothing to share: one ordinary launch
        for mode in (0, 2):
            lib.rdf_set_scheduler(mode)
            out = outs[2].fill(65535)
            with torch.cuda.stream(main):
                assert ev.get_labels_forest_split(f, depth, out, helper.cuda_stream, 32) == 0
            torch.cuda.synchronize()
            assert np.array_equal(out.get(), want), mode
        lib.rdf_set_scheduler(-1)
        # helper_cus 0, or the same stream twice: one ordinary launch
        for hs, cus in ((helper.cuda_stream, 0), (main.cuda_stream, 32)):
            out = outs[1].fill(65535)
            with torch.cuda.stream(main):
                assert ev.get_labels_forest_split(f, depth, out, hs, cus) == 0
            torch.cuda.synchronize()
            assert np.array_equal(out.get(), want)
    finally:
        torch.cuda.synchronize()
        lib.rdf_set_scheduler(-1)
        lib.rdf_stream_destroy(h)


def test_first_evaluation_of_an_unseen_table_cannot_be_captured(rdf, gpu_runtime):
    """RDF_ERR_CAPTURE, not a generic bad argument: the first evaluation of a table the process has not seen reads its info
    block back, which a stream under capture cannot do; after one look at the table (an evaluation, a pack, rdf_forest_info --
    which is what adopt_packed does at once since round 6) capture works."""
    import torch
    lib = gpu_runtime.lib
    forest_np = rdf.synth.forest(2, 8, 4, "trained", 3)
    f = rdf.DecisionForest.from_numpy(forest_np)
    table = rdf.DecisionForest.from_numpy(forest_np).packed_bytes(1.0)
    raw = rdf.DeviceArray((table.size,), np.uint8).set(table)       # a plain upload: the library has not looked at these bytes
    depth = rdf.to_device(rdf.synth.frames(["dense"], 1, 64, 96))
    out = rdf.DeviceArray((1, 64, 96), np.uint16).fill(65535)
    side = torch.cuda.Stream()

    def captured(ptr):
        graph = torch.cuda.CUDAGraph()
        rcs = []
        with torch.cuda.graph(graph, stream=side):
            out.fill(65535)         # (also: something to record -- an empty capture is an error of its own on some runtimes)
            rcs.append(lib.rdf_eval_forest_packed(depth.ptr, 1, 96, 64, ptr, f.forest_cu.ptr, 2, 8, 4, None, -1, out.ptr, 1,
                                                  gpu_runtime.stream()))
        return rcs, graph

    rcs, _ = captured(raw.ptr)
    assert rcs == [-6] and b"captured" in lib.rdf_error_string(-6)
    assert lib.rdf_eval_forest_packed(depth.ptr, 1, 96, 64, raw.ptr, f.forest_cu.ptr, 2, 8, 4, None, -1, out.ptr, 1,
                                      gpu_runtime.stream()) == 0
    want = out.get()
    # an ADOPTED table has been looked at: its very first evaluation can be recorded
    f.adopt_packed(table, 1.0)
    rcs, graph = captured(f.packed(1.0).ptr)
    assert rcs == [0]
    out.fill(0)
    graph.replay()
    torch.cuda.synchronize()
    assert np.array_equal(out.get(), want) and (want != 65535).any()
    lib.rdf_forest_forget(raw.ptr)


def test_deep_blocks_at_config5_size(rdf, evs, oracle, gpu_runtime):
    """BASELINE config 5's forest shape on a forest whose deep levels are occupied: T8/D22/C4 balanced (3.6 GiB of packed
    tables, 1.2 GB of them deep blocks: block offsets up to 2^29 inside a root level), two dense 1280x720 frames, the walk
    taking over at root level 11 (what DecisionForest.tune picks on config 5's shard) and at 20 (the last block alone), and the
    heap-order records: the oracle's labels every time.  (tree_eval.cu:95-128 is what all three compute.)"""
    lib = gpu_runtime.lib
    synth = rdf.synth
    forest = synth.forest(8, 22, 4, "balanced", calib=synth.frames(["dense"] * 4, 9000, 720, 1280))
    depth = synth.frames(["dense", "dense"], 5000, 720, 1280)
    want = np.full(depth.shape, 65535, np.uint16)
    st = np.zeros(3, np.uint64)
    oracle.eval_forest(depth, forest, want, stats=st)
    px = int(st[0])
    assert px == depth.size and int(st[1]) == px * 8 * 22 and int(st[2]) == px * 8      # every walk reaches level D-1
    f = rdf.DecisionForest.from_numpy(forest)
    d = rdf.to_device(depth)
    try:
        for level in (11, 20, 0):
            lib.rdf_set_deep_from(level)
            out = rdf.DeviceArray(depth.shape, np.uint16).fill(65535)
            evs["packed"].get_labels_forest(f, d, out)
            got = out.get()
            assert np.array_equal(got, want), (level, int((got != want).sum()))
            blocks = _blocks_walked(rdf, gpu_runtime, f, d, depth.shape)
            assert blocks == (px * 8 * (len(range(level, 20, 3)) + 1) if level else 0), (level, blocks)
    finally:
        lib.rdf_set_deep_from(-1)


def _spine_forest(T, D, C, spine_trees):
    """A forest of 2^D - 1 nodes per tree that costs no host memory to speak of: all zeros (numpy's zeros are untouched
    pages) except the leftmost node of every level in `spine_trees`.  Such a node compares depth[x] - 65535 (the v probe lies
    262 columns to the right: outside a 160-column frame) with a threshold that falls from level to level, so the lanes of a
    wave leave the walk a few per level (to the right: a leaf with a one-hot PDF) and the rest goes on to the left, down to
    level D - 1.  All-zero trees end at their root (tree_eval.cu:107-121: 0 < 0 is false, the right side is a leaf of zeros)."""
    f = np.zeros((T, (1 << D) - 1, 7 + 2 * C), np.float32)
    rows = {}
    for k in spine_trees:
        for j in range(D):
            row = np.zeros(7 + 2 * C, np.float32)
            row[2] = float(1 << 20)                                   # v.x: an integer-record numerator (|a| < 2^21)
            row[4] = float((4400 - 30 * j - 7 * k) - 65535)           # left iff depth < 4400 - 30 j - 7 k
            row[5] = -1.0 if j < D - 1 else 0.0                       # left: continue (a leaf on the last level)
            row[6] = 0.0                                              # right: leaf
            row[7 + (j + k) % C] = (1 + (j % 3)) / 4.0                # left PDF (used on the last level only)
            row[7 + C + (j + 2 * k + 1) % C] = (2 + (j % 5)) / 8.0    # right PDF
            f[k, (1 << j) - 1] = row
            rows[(k, (1 << j) - 1)] = row
    return f, rows


@pytest.mark.parametrize("T", [6, 8])
def test_depth_24_forests_walk_their_deep_blocks(rdf, oracle, gpu_runtime, T):
    """The deep walk addresses a block as 64-bit base + 32-bit offset (heap index x 128): the deepest forests that get deep
    blocks, T6 and T8 at D24 (3.7 and 4.9 GB of blocks: block offsets up to 2^30, table offsets beyond 2^32 -- round 4 pointed
    finished lanes at a zero line behind the last block with a 32-bit difference that wrapped for T8/D24; now a finished lane
    names the first block of its tree's level).  Every wave holds finished and walking lanes side by side.  Labels are the oracle's."""
    deep = True
    lib = gpu_runtime.lib
    D, C = 24, 4
    with_deep = lib.rdf_forest_packed_bytes(T, D, C)
    slots_and_last = ((T << D) * (16 + 32) + (T << (D - 1)) * 64 + 64 + 127) & ~127
    assert (with_deep > slots_and_last + 128) == deep
    forest_np, rows = _spine_forest(T, D, C, spine_trees=(0, 2, T - 1))
    depth_np = rdf.synth.frames(["dense", "dense", "live"], 4300, 96, 160)
    want = np.full(depth_np.shape, 65535, np.uint16)
    st = np.zeros(3, np.uint64)
    oracle.eval_forest(depth_np, forest_np, want, stats=st)
    assert int(st[1]) > 6 * int(st[0]) and len(np.unique(want)) >= 4          # walks of many lengths, several classes
    f = rdf.DecisionForest(T, D, C)                                  # zeros on the device; the spine nodes one by one
    for (k, node), row in rows.items():
        f.forest_cu[k][node].set(row)
    depth = rdf.to_device(depth_np)
    ev = rdf.DecisionTreeEvaluator()
    ev.auto_tune = False
    try:
        for level in (19, 22, 10, 0):
            lib.rdf_set_deep_from(level)
            for block in (256, 512):
                lib.rdf_set_block_threads(block)
                out = rdf.DeviceArray(depth_np.shape, np.uint16).fill(65535)
                ev.get_labels_forest(f, depth, out)
                got = out.get()
                assert np.array_equal(got, want), (T, level, block, int((got != want).sum()))
            lib.rdf_set_block_threads(0)
            blocks = _blocks_walked(rdf, gpu_runtime, f, depth, depth_np.shape)
            assert (blocks > 0) == (deep and level > 0), (T, level, blocks)
    finally:
        lib.rdf_set_deep_from(-1)
        lib.rdf_set_block_threads(0)


@pytest.mark.parametrize("T,D,C", [(2, 25, 4), (1, 27, 4), (1, 28, 2), (1, 30, 1)])
def test_the_deepest_forests_the_abi_accepts(rdf, oracle, gpu_runtime, T, D, C):
    """The ABI takes max_depth <= 30, packed tables <= 27, deep blocks <= 24; until round 6 nothing deeper than 24 had run on
    hardware.  Spine forests (all zeros but one path per tree: no host memory to speak of) at D25 and D27 through the packed
    tables -- 32-bit byte offsets inside one tree up to 2^31 -- and at D28 and D30 through the reference layout, where the
    reference's own int32 addressing (cu_utils.hpp:32-39: (idx_offset + node) * els_per_node) has long wrapped and this
    library's 64-bit addressing has not: the oracle's labels, walks that end on every level down to D - 1."""
    try:        # (the forest is 2^D x (7 + 2C) floats of untouched zero pages on the host -- up to 39 GB of address space at D30 -- and as much
                # real memory on the device: a box that cannot give either skips the case instead of failing the suite)
        forest_np, rows = _spine_forest(T, D, C, spine_trees=tuple(range(T)))
    except MemoryError:
        pytest.skip(f"no {((T << D) * (7 + 2 * C) * 4) >> 30} GiB of host address space for a D{D} forest on this box")
    depth_np = rdf.synth.frames(["dense", "dense", "live"], 4300, 96, 160)
    want = np.full(depth_np.shape, 65535, np.uint16)
    st = np.zeros(3, np.uint64)
    oracle.eval_forest(depth_np, forest_np, want, stats=st)
    lengths = oracle.walk_lengths(depth_np, forest_np)
    assert int(lengths.max()) == D and int(st[1]) > 6 * int(st[0])                 # some walk goes all the way down
    del forest_np
    import torch
    if torch.cuda.mem_get_info()[0] < 2.6 * ((T << D) * (7 + 2 * C) * 4):           # the forest, and up to 1.4 x as much of packed tables
        pytest.skip("not enough free device memory for this depth")
    f = rdf.DecisionForest(T, D, C)                                  # zeros on the device; the spine nodes one by one
    for (k, node), row in rows.items():
        f.forest_cu[k][node].set(row)
    depth = rdf.to_device(depth_np)
    ev = rdf.DecisionTreeEvaluator()
    ev.auto_tune = False
    lib = gpu_runtime.lib
    assert (lib.rdf_forest_packed_bytes(T, D, C) > 0) and ((D <= 27) == (f.max_depth <= 27))
    try:
        for block in (256, 512):
            lib.rdf_set_block_threads(block)
            out = rdf.DeviceArray(depth_np.shape, np.uint16).fill(65535)
            ev.get_labels_forest(f, depth, out)                      # packed for D <= 27, the reference layout beyond
            got = out.get()
            assert np.array_equal(got, want), (T, D, C, block, int((got != want).sum()))
        if D <= 27:     # ... and a packed forest's reference-layout evaluation, for the same answer
            lib.rdf_set_block_threads(0)
            out = rdf.DeviceArray(depth_np.shape, np.uint16).fill(65535)
            rdf.DecisionTreeEvaluator(use_packed=False).get_labels_forest(f, depth, out)
            assert np.array_equal(out.get(), want)
        else:
            assert lib.rdf_forest_pack(f.forest_cu.ptr, T, D, C, 1.0, f.forest_cu.ptr, gpu_runtime.stream()) == -1      # RDF_ERR_BAD_ARG: packed tables stop at 27
    finally:
        lib.rdf_set_block_threads(0)
    del f
    import torch
    torch.cuda.empty_cache()


def test_deep_blocks_fuzz(rdf, evs, oracle, gpu_runtime):
    """Seeded fuzz of the deep-block walk: random trees (1-9), depths (5-15), classes (1-8), topologies (balanced, trained-like,
    full; a few wild nodes), frames, labels_reduce, scale, filter, pre-fill, take-over level and launch geometry -- every
    combination bit-exact against the C oracle.  RDF_DEEP_FUZZ_ROUNDS / RDF_FUZZ_SEED for soaks (profiles/r04_fuzz_soak.log)."""
    rounds = int(os.environ.get("RDF_DEEP_FUZZ_ROUNDS", "40"))
    rng = np.random.default_rng(int(os.environ.get("RDF_FUZZ_SEED", "20261004")) + 17)
    lib = gpu_runtime.lib
    calib = rdf.synth.calibration_frames(3, 96, 160)
    try:
        for it in range(rounds):
            T, D, C = int(rng.integers(1, 10)), int(rng.integers(5, 16)), int(rng.integers(1, 9))
            topology = str(rng.choice(["balanced", "balanced", "trained", "full"]))
            forest = (rdf.synth.forest(T, D, C, "balanced", first_tree=it, calib=calib) if topology == "balanced"
                      else rdf.synth.forest(T, D, C, topology, first_tree=it))
            if rng.random() < 0.25:     # wild nodes somewhere in the upper half of the levels: blocks below them stay usable
                lvl = int(rng.integers(0, max(1, D // 2)))
                node = (1 << lvl) - 1 + int(rng.integers(0, 1 << lvl))
                forest[int(rng.integers(0, T)), node, int(rng.integers(0, 4))] = rng.choice([3e7, np.inf, np.nan, 1e-41])
            n, h, w = int(rng.integers(1, 4)), int(rng.integers(8, 120)), int(rng.integers(8, 260))
            if rng.random() < 0.5:
                w = max(8, w & ~7)
            r = int(rng.choice([1, 1, 2, 3]))
            s = float(rng.choice([1.0, 0.5, 1.5]))
            depth = rdf.synth.frames([str(k) for k in rng.choice(["dense", "live"], size=n)], 8000 + it, h, w)
            use_filter = rng.random() < 0.3
            filt = rng.integers(0, 3, size=(n, h // r, w // r)).astype(np.uint16) if use_filter else None
            prefill = int(rng.choice([65535, 0]))
            last = 2 if C <= 4 else 1
            roots = list(range((D - last) % 3, D - last + 1, 3))
            lib.rdf_set_deep_from(int(rng.choice(roots + [1, D])))
            lib.rdf_set_block_threads(int(rng.choice([0, 256, 512])))
            lib.rdf_set_halo(int(rng.choice([-1, -1, 8, 40])))
            lib.rdf_set_lds_levels(int(rng.choice([-1, -1, 0, 3])))
            lib.rdf_set_rows_per_wave(int(rng.choice([0, 1, 2, 4])))
            lib.rdf_set_group(int(rng.choice([0, 0, 2, 3, 4])))
            lib.rdf_set_scheduler(int(rng.choice([-1, 0, 1, 2])))
            lib.rdf_set_tree_waves(0)
            want = np.full((n, h // r, w // r), prefill, np.uint16)
            oracle.eval_forest(depth, forest, want, r, filt, 2 if use_filter else None, s)
            got = _gpu_forest(rdf, evs["packed"], depth, forest, prefill, r, filt, 2 if use_filter else None, s)
            assert np.array_equal(got, want), f"iteration {it}: {topology} T{T} D{D} C{C} {n}x{h}x{w} r{r} s{s} filter {use_filter}"
    finally:
        for knob, v in (("deep_from", -1), ("block_threads", 0), ("halo", -1), ("lds_levels", -1), ("rows_per_wave", 0), ("group", 0),
                        ("scheduler", -1), ("tree_waves", -1)):
            getattr(lib, "rdf_set_" + knob)(v)
