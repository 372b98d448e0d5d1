"""training of one tree.  CPU: the numpy restatement learns a separable toy problem.
GPU (-m gpu): DecisionTreeTrainer on the device produces the SAME tree, bit for bit, as the restatement
fed with the same proposals (counts are integers; the fp32 gain arithmetic is evaluated as written)."""
import importlib

import numpy as np
import pytest

from oracle import train_numpy as tn


def make_data(rdf, n=6, h=40, w=56, first=50):
    depth = rdf.synth.frames(["live"] * n, first, h, w)
    labels = np.zeros((n, h, w), np.uint16)
    yy, xx = np.mgrid[0:h, 0:w]
    for i in range(n):
        valid = (depth[i] != 0) & (depth[i] != 65535)
        cls = 1 + ((xx > w // 2).astype(int) + 2 * (depth[i] > 4000).astype(int)) % 3
        labels[i][valid] = cls[valid]
    labels[0, 0, 0] = 2              # a labelled pixel on a missing-depth cell (d = 65535): still counted
    depth[0, 0, 0] = 65535
    labels[0, 1, 1] = 1              # and one with depth 0: compute_feature returns 0.f
    depth[0, 1, 1] = 0
    return depth, labels


class _ArrayDataset:
    """Minimal stand-in for DecisionTreeDatasetConfig over in-memory arrays."""

    def __init__(self, depth, labels, n_classes, per_block):
        self.depth, self.labels, self._c = depth, labels, n_classes
        self.num_images, self.images_per_block = depth.shape[0], per_block
        self.img_dims = (depth.shape[2], depth.shape[1])

    def num_classes(self):
        return self._c

    def images_shape(self):
        return self.depth.shape

    def get_depth_block_cu(self, b, arr):
        arr.set(self.depth[b * self.images_per_block:(b + 1) * self.images_per_block])

    def get_labels_block_cu(self, b, arr):
        arr.set(self.labels[b * self.images_per_block:(b + 1) * self.images_per_block])


def test_numpy_trainer_learns_a_separable_problem(rdf, oracle):
    depth, labels = make_data(rdf)
    np.random.seed(42)
    tree = tn.train_tree(depth, labels, 4, 7, 2, 24)
    out = np.full(depth.shape, 65535, np.uint16)
    oracle.eval_tree(depth, tree, out)
    m = labels > 0
    assert (out[m] == labels[m]).mean() > 0.6
    # record format invariants (tree_train.cu:183-235)
    flags = tree[:, 5:7]
    assert set(np.unique(flags)).issubset({-1.0, 0.0})
    last = tree[(1 << 6) - 1:]
    assert (last[:, 5:7] == 0).all()


@pytest.mark.gpu
@pytest.mark.parametrize("sorted_rows", [True, False], ids=["sorted_rows", "wave_atomics"])
@pytest.mark.parametrize("cfg", [(6, 2, 16, 1 << 17), (5, 1, 40, 1 << 17), (7, 3, 8, 8)],
                         ids=["D6_2x16", "D5_1x40", "D7_3x8_nodeblocks8"])
def test_device_trainer_matches_restatement_bit_for_bit(cfg, sorted_rows, rdf, gpu_runtime):
    """Both ways of counting -- rows of decision bits in (node, class) order (the default) and the histogram kernel with
    its per-wave atomics -- train the restatement's tree, bit for bit."""
    D, blocks, P, max_nodes = cfg
    depth, labels = make_data(rdf, n=6)
    C = 4
    ds = _ArrayDataset(depth, labels, C, per_block=3)
    trainer = rdf.DecisionTreeTrainer(3, P)
    trainer.MAX_NEXT_NODES_TO_COUNT_PER_BLOCK = max_nodes
    trainer.allocate(ds, blocks * P, D)
    assert trainer.use_sorted_rows
    trainer.use_sorted_rows = sorted_rows
    trainer.SORTED_ROWS_FROM_ACTIVE_NODES = 1      # (every level by sorted rows, the root included)
    tree = rdf.DecisionTree(D, C)
    np.random.seed(7)
    trainer.train(ds, tree)
    got = tree.tree_out_cu.get()

    def proposals(n):
        arr = np.zeros((n, 5), np.float32)
        importlib.import_module("3d-beats_amd.decision_tree").make_random_features(n, arr)
        return arr

    np.random.seed(7)
    want = tn.train_tree(depth, labels, C, D, blocks, P, max_next_nodes_per_block=max_nodes, proposal_fn=proposals)
    assert got.shape == want.shape
    same = got.view(np.uint32) == want.view(np.uint32)
    assert same.all(), f"{(~same).sum()} of {same.size} words differ; first at {np.argwhere(~same)[:5].tolist()}"
    # and it is reproducible run to run
    np.random.seed(7)
    trainer.train(ds, tree)
    assert np.array_equal(tree.tree_out_cu.get().view(np.uint32), got.view(np.uint32))


@pytest.mark.gpu
def test_histogram_variants_agree_when_big_and_small_bins_mix(rdf, gpu_runtime):
    """At one level, (node, class) bins above 65535 pixels (32-bit pair counters) next to bins below (16-bit quad
    counters): the workspace variant must still equal the plain one-atomic-per-group kernel."""
    dt = importlib.import_module("3d-beats_amd.decision_tree")
    n, h, w, C, D, P = 6, 240, 424, 4, 3, 37
    depth = rdf.synth.frames(["dense"] * n, 7100, h, w)
    rng = np.random.default_rng(3)
    labels = np.where(rng.random(depth.shape) < 0.9, 1, rng.integers(2, C, size=depth.shape)).astype(np.uint16)
    ds = _ArrayDataset(depth, labels, C, per_block=n)
    trainer = rdf.DecisionTreeTrainer(n, P)
    trainer.allocate(ds, P, D)
    tree = rdf.DecisionTree(D, C)
    np.random.seed(21)
    trainer.train(ds, tree)
    lib, st = gpu_runtime.lib, gpu_runtime.stream
    parents = trainer.node_counts_cu.get().reshape(-1, C)
    nodes_px = trainer.nodes_by_pixel_cu.get()
    live_nodes = np.unique(nodes_px[nodes_px >= 0])
    sizes = parents[live_nodes]
    assert (sizes > 65535).any() and ((sizes > 0) & (sizes <= 65535)).any(), sizes
    props = np.zeros((P, 5), np.float32)
    np.random.seed(8)
    dt.make_random_features(P, props)
    d_props = rdf.to_device(props)
    NB = trainer.nodes_per_block
    args = (trainer.depth_cu.ptr, trainer.labels_cu.ptr, trainer.nodes_by_pixel_cu.ptr, n, w, h, d_props.ptr, P, C, 0, 1 << D, NB)
    plain = rdf.DeviceArray((P, NB, C), np.uint64).fill(0)
    assert lib.rdf_train_histogram_left(*args, plain.ptr, st()) == 0
    ws = rdf.DeviceArray((int(lib.rdf_train_histogram_workspace_bytes(P, NB, C)),), np.uint8).fill(0)
    for parents_ptr in (trainer.node_counts_cu.ptr, None):
        fast = rdf.DeviceArray((P, NB, C), np.uint64).fill(0)
        assert lib.rdf_train_histogram_left_ws(*args, fast.ptr, ws.ptr, parents_ptr, st()) == 0
        assert np.array_equal(fast.get(), plain.get())
        assert not ws.get().any()
    assert plain.get().sum() > 0


@pytest.mark.gpu
@pytest.mark.parametrize("P", [37, 64, 200, 1024])
def test_sorted_row_counts_equal_the_histogram_kernels(P, rdf, gpu_runtime):
    """rdf_train_sort_pixels + rdf_train_decision_bits + rdf_train_count_rows leave the count array exactly as
    rdf_train_histogram_left does: on the deepest level of a trained tree (many small groups), in node blocks, with pixels
    whose label is beyond the class count, and for proposal counts that are no power of two."""
    import ctypes
    dt = importlib.import_module("3d-beats_amd.decision_tree")
    depth, labels = make_data(rdf, n=6)
    labels = labels.copy()
    labels[0, :4, :9] = 9                      # labels >= C are ignored by both paths
    C, D = 4, 7
    ds = _ArrayDataset(depth, labels, C, per_block=6)
    trainer = rdf.DecisionTreeTrainer(6, 24)
    trainer.allocate(ds, 24, D)
    tree = rdf.DecisionTree(D, C)
    np.random.seed(31)
    trainer.train(ds, tree)                    # leaves nodes_by_pixel at level D - 1
    lib, st = gpu_runtime.lib, gpu_runtime.stream
    n, h, w = depth.shape
    level = D - 1
    n_nodes = 2 ** level
    nodes_px = trainer.nodes_by_pixel_cu.get()
    assert len(np.unique(nodes_px[nodes_px >= 0])) > 8
    props = np.zeros((P, 5), np.float32)
    np.random.seed(32)
    dt.make_random_features(P, props)
    d_props = rdf.to_device(props)
    row_bytes = int(lib.rdf_train_bits_row_bytes(P))
    assert row_bytes == {37: 8, 64: 8, 200: 32, 1024: 128}[P]
    n_lab = int((labels != 0).sum())
    pos = rdf.DeviceArray(depth.shape, np.int32)
    rowkey = rdf.DeviceArray((n_lab,), np.int32)
    bits = rdf.DeviceArray((n_lab * row_bytes,), np.uint8)
    work = rdf.DeviceArray((int(lib.rdf_train_sort_workspace_bytes(n_nodes, C)),), np.uint8).fill(255)   # (the call zeroes it)
    assert lib.rdf_train_sort_pixels(trainer.labels_cu.ptr, trainer.nodes_by_pixel_cu.ptr, depth.size, C, n_nodes, pos.ptr,
                                     rowkey.ptr, work.ptr, st()) == 0
    # the rows are a permutation of the live pixels, grouped by (node, class) in ascending order
    lab = labels
    live = (nodes_px >= 0) & (lab < C)
    got_pos = pos.get()
    assert (got_pos[~live] == -1).all()
    rows = got_pos[live]
    assert sorted(rows.tolist()) == list(range(int(live.sum())))
    keys = rowkey.get()[:int(live.sum())]
    assert (np.diff(keys) >= 0).all()
    assert np.array_equal(keys[rows], (nodes_px[live] * C + lab[live]).astype(np.int32))
    bws = rdf.DeviceArray((int(lib.rdf_train_bits_workspace_bytes(P)),), np.uint8)
    assert lib.rdf_train_decision_bits(trainer.depth_cu.ptr, pos.ptr, n, w, h, d_props.ptr, P, bits.ptr, bws.ptr, st()) == 0
    for NB in (2 * n_nodes, 16):               # one node block, then blocks of 16 next-level nodes
        for start in range(0, 2 * n_nodes, NB):
            end = start + NB
            want = rdf.DeviceArray((P, NB, C), np.uint64).fill(0)
            assert lib.rdf_train_histogram_left(trainer.depth_cu.ptr, trainer.labels_cu.ptr, trainer.nodes_by_pixel_cu.ptr, n, w, h,
                                                d_props.ptr, P, C, start, end, NB, want.ptr, st()) == 0
            got = rdf.DeviceArray((P, NB, C), np.uint64).fill(0)
            assert lib.rdf_train_count_rows(bits.ptr, rowkey.ptr, work.ptr, n_nodes, P, C, start, end, NB, got.ptr, st()) == 0
            assert np.array_equal(got.get(), want.get()), (P, NB, start)
            if NB == 2 * n_nodes:
                assert want.get().sum() > 0


@pytest.mark.gpu
def test_left_only_histogram_plus_right_counts_equals_full_histogram(rdf, gpu_runtime):
    """rdf_train_histogram_left + rdf_train_right_counts must leave the count array exactly as the one-call
    rdf_train_histogram (= evaluate_random_features) does, here on the last level of a trained tree (many nodes)."""
    dt = importlib.import_module("3d-beats_amd.decision_tree")
    depth, labels = make_data(rdf, n=6)
    C, D, P = 4, 6, 24
    ds = _ArrayDataset(depth, labels, C, per_block=6)
    trainer = rdf.DecisionTreeTrainer(6, P)
    trainer.allocate(ds, P, D)
    tree = rdf.DecisionTree(D, C)
    np.random.seed(11)
    trainer.train(ds, tree)
    lib, st = gpu_runtime.lib, gpu_runtime.stream
    # state left behind by train(): nodes_by_pixel / active nodes / node counts of level D-1
    active = trainer.active_nodes_cu.get()
    counts_parent = trainer.node_counts_cu.get().reshape(-1, C)
    nodes_px = trainer.nodes_by_pixel_cu.get()
    live_nodes = np.unique(nodes_px[nodes_px >= 0])
    assert live_nodes.size > 4
    n_act = live_nodes.size
    assert np.array_equal(np.sort(active[:n_act]), live_nodes)
    props = np.zeros((P, 5), np.float32)
    np.random.seed(5)
    dt.make_random_features(P, props)
    d_props = rdf.to_device(props)
    NB = trainer.nodes_per_block
    n_img, h, w = depth.shape
    start, end = 0, 1 << D
    assert end - start <= NB
    full = rdf.DeviceArray((P, NB, C), np.uint64).fill(0)
    split = rdf.DeviceArray((P, NB, C), np.uint64).fill(0)
    args = (trainer.depth_cu.ptr, trainer.labels_cu.ptr, trainer.nodes_by_pixel_cu.ptr, n_img, w, h, d_props.ptr, P, C,
            start, end, NB)
    assert lib.rdf_train_histogram(*args, full.ptr, st()) == 0
    assert lib.rdf_train_histogram_left(*args, split.ptr, st()) == 0
    left_only = split.get()
    assert lib.rdf_train_right_counts(n_act, trainer.active_nodes_cu.ptr, P, NB, start, end, C, trainer.node_counts_cu.ptr,
                                      split.ptr, st()) == 0
    # ... and the workspace variant (two proposals per 64-bit atomic) leaves the same left counts and a zero workspace
    ws_bytes = int(lib.rdf_train_histogram_workspace_bytes(P, NB, C))
    assert ws_bytes == NB * C * ((P + 3) // 4 * 4) * 6
    ws = rdf.DeviceArray((ws_bytes,), np.uint8).fill(0)
    packed = rdf.DeviceArray((P, NB, C), np.uint64).fill(0)
    for parents in (None, trainer.node_counts_cu.ptr, None, trainer.node_counts_cu.ptr):
        # twice each: a call relies on the previous one having cleaned up; with the parents' counts the small
        # (node, class) bins -- all of them in this little problem -- count four proposals per 64-bit word
        packed.fill(0)
        assert lib.rdf_train_histogram_left_ws(*args, packed.ptr, ws.ptr, parents, st()) == 0
        assert np.array_equal(packed.get(), left_only)
        assert not ws.get().any()
    want, got = full.get(), split.get()
    assert want.sum() == P * (nodes_px >= 0).sum()
    assert (left_only[:, 1::2, :] == 0).all() and left_only.sum() < want.sum()
    assert np.array_equal(got, want)
    # parents' counts are what the children add up to
    for node in live_nodes[:5]:
        assert np.array_equal(want[0, 2 * node] + want[0, 2 * node + 1], counts_parent[node])


@pytest.mark.parametrize("seed,n", [(123, 9), (0, 1), (7, 2000), (20211003, 257)])
def test_proposal_generator_matches_restatement(seed, n, rdf):
    """The bulk generator (11 raw MT19937 outputs per proposal) must give the restatement's per-draw proposals
    bit for bit AND leave the global RNG in the same state, so that later draws stay aligned too."""
    dt = importlib.import_module("3d-beats_amd.decision_tree")
    np.random.seed(seed)
    a = np.zeros((n, 5), np.float32)
    dt.make_random_features(n, a)
    state_a = np.random.get_state()
    np.random.seed(seed)
    b = tn.make_random_features(n)
    state_b = np.random.get_state()
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    assert np.array_equal(state_a[1], state_b[1]) and state_a[2:] == state_b[2:]
    np.random.seed(seed)
    c = np.zeros((n, 5), np.float32)
    dt.make_random_features_loop(n, c)
    assert np.array_equal(a.view(np.uint32), c.view(np.uint32))


@pytest.mark.gpu
def test_train_forest_flow_from_a_dataset_directory(rdf, gpu_runtime, tmp_path):
    """train_model.py's flow end to end: dataset directory -> candidate trees -> best trees -> forest .npy that the
    inference path loads and that beats chance on held-out frames."""
    ds_mod = importlib.import_module("3d-beats_amd.dataset")
    depth, labels = make_data(rdf, n=10, h=48, w=64, first=300)
    ddir = tmp_path / "data"
    ds_mod.write_dataset(str(ddir), depth, labels, {1: [255, 0, 0, 255], 2: [0, 255, 0, 255], 3: [0, 0, 255, 255]})
    np.random.seed(5)
    out = tmp_path / "forest.npy"
    forest_cpu, pct = ds_mod.train_forest(str(ddir), 6, 4, 32, 16, 2, 6, str(out), trees_to_try=3, log=lambda *_: None)
    assert forest_cpu.shape == (2, 63, 15) and out.exists()
    assert pct > 0.5
    f = rdf.DecisionForest.load(str(out))
    assert (f.num_trees, f.max_depth, f.num_classes) == (2, 6, 4)


@pytest.mark.gpu
def test_device_trainer_fuzz_against_restatement(rdf, gpu_runtime):
    """Seeded random training problems (image sizes that do not divide the 32x8 tiles, odd proposal counts, node
    blocks, 2-6 classes, unlabelled and invalid-depth pixels): the trained tree must equal the restatement's bit for
    bit.  RDF_TRAIN_FUZZ_ROUNDS / RDF_TRAIN_FUZZ_SEED for longer soaks."""
    import os
    dt = importlib.import_module("3d-beats_amd.decision_tree")
    rounds = int(os.environ.get("RDF_TRAIN_FUZZ_ROUNDS", "4"))
    rng = np.random.default_rng(int(os.environ.get("RDF_TRAIN_FUZZ_SEED", "4242")))
    for it in range(rounds):
        n, h, w = int(rng.integers(1, 5)), int(rng.integers(9, 70)), int(rng.integers(9, 90))
        C, D = int(rng.integers(2, 7)), int(rng.integers(2, 7))
        P, blocks = int(rng.integers(1, 23)), int(rng.integers(1, 3))
        max_nodes = int(rng.choice([1 << 17, 8, 4]))
        depth = rdf.synth.frames([str(k) for k in rng.choice(["dense", "live"], size=n)], 9000 + it, h, w)
        depth[rng.random(depth.shape) < 0.03] = 0
        labels = rng.integers(0, C, size=depth.shape).astype(np.uint16)
        if rng.random() < 0.5:                      # spatially coherent labels instead of noise
            yy, xx = np.mgrid[0:h, 0:w]
            labels[:] = (1 + ((xx * 3 // max(w, 1)) + (depth > 4000)) % (C - 1)).astype(np.uint16)
        labels[rng.random(depth.shape) < 0.3] = 0   # unlabelled
        if (labels > 0).sum() == 0:
            labels[0, 0, 0] = 1
        ds = _ArrayDataset(depth, labels, C, per_block=n)
        trainer = rdf.DecisionTreeTrainer(n, P)
        trainer.MAX_NEXT_NODES_TO_COUNT_PER_BLOCK = max_nodes
        trainer.allocate(ds, blocks * P, D)
        # both ways of counting: rows of decision bits from the root on, from 4 or 16 active nodes on, or never
        trainer.SORTED_ROWS_FROM_ACTIVE_NODES = int(rng.choice([1, 1, 4, 16]))
        trainer.use_sorted_rows = bool(rng.random() < 0.8)
        tree = rdf.DecisionTree(D, C)
        seed = int(rng.integers(0, 2 ** 31))
        np.random.seed(seed)
        trainer.train(ds, tree)
        got = tree.tree_out_cu.get()

        def proposals(k):
            arr = np.zeros((k, 5), np.float32)
            dt.make_random_features(k, arr)
            return arr

        np.random.seed(seed)
        want = tn.train_tree(depth, labels, C, D, blocks, P, max_next_nodes_per_block=max_nodes, proposal_fn=proposals)
        same = got.view(np.uint32) == want.view(np.uint32)
        assert same.all(), (f"round {it}: n{n} {h}x{w} C{C} D{D} P{P}x{blocks} nodes/block {max_nodes}: "
                            f"{(~same).sum()} words differ, first at {np.argwhere(~same)[:3].tolist()}")
