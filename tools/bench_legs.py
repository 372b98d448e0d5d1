"""The legs of bench.py that time the rows either side of the forest kernel: the app's per-hand chain,
mean shift, training of one tree.  Every leg checks what it timed against the CPU restatements (oracle/: test
infrastructure, used here only as the checker) on a bounded sample and returns a plain dict for the bench line.
tests/perf/*.py are the stand-alone command lines of the same functions.
"""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def _timed(fn, n, sync):
    sync()
    t0 = time.perf_counter()
    for i in range(n):
        fn(i)
    sync()
    return (time.perf_counter() - t0) / n


def hand_pipeline(rdf, n=200, check=True):
    """Per-hand, per-frame latency of the app's chain (3d_bz.py:388-522) through HandPipeline: stencil, flip, 0->65535,
    2-layer forest (labels_reduce 2), flip back, RGBA, 6 mean-shift rounds, fingertip heights, the result on the host.
    848x480 frame with two synthetic hands; forests of the size the reference's models have (4 trees, depth 18)."""
    import torch
    from test_pipeline import H, W, R, _scene
    pl = importlib.import_module("3d-beats_amd.pipeline")
    depth, groups = _scene(rdf)
    f0, f1 = rdf.synth.forest(4, 18, 4, "trained", 60), rdf.synth.forest(4, 18, 5, "trained", 70)
    conditions = [[0, 1], [0, 2], [1, 3], [0, 3], [0, 4], [0, 5], [0, 6], [0, 7]]
    colors = [[10 * i, 255 - 10 * i, i, 255] for i in range(1, 8)]
    cfg = {"layers": [{"model": rdf.DecisionForest.from_numpy(f0)},
                      {"model": rdf.DecisionForest.from_numpy(f1), "filter_model": 0, "filter_model_class": 3}],
           "conditions": conditions, "label_colors": colors}
    lf = rdf.LayeredDecisionForest(cfg, (H, W), R)
    variances, tips, intr, plane = np.full(7, 40., np.float32), [3, 4, 5, 6, 7], (421.3, 420.9, 423.1, 238.6), np.eye(4, dtype=np.float32)
    args = ((H, W), R, 1.0, 6, variances, tips, intr, plane)
    pipe = pl.HandPipeline(lf, *args)
    dbuf, gbuf = rdf.GpuBuffer((H, W), np.uint16), rdf.GpuBuffer((H, W), np.uint16)
    dbuf.cu().set(depth)
    gbuf.cu().set(groups)
    sync = torch.cuda.synchronize
    out = {"frame": [H, W], "labels_reduce": R, "layers": 2, "trees": 4, "tree_depth": 18, "mean_shift_rounds": 6,
           "fingertips": len(tips)}

    if check:     # one hand of the frame against the same chain built from the restatements
        from oracle import mean_shift_numpy as ms_np
        from oracle import points_ops_numpy as po_np
        from oracle import rdf_oracle
        means, heights = pipe.run(dbuf, gbuf, 1, True)
        d_group = po_np.stencil_depth_image_by_group(W, H, 0, 1, groups, depth, np.zeros((H, W), np.uint16))
        d2 = d_group[:, ::-1].copy()
        po_np.convert_0s_to_maxuint(d2)
        l0 = np.full((1, H // R, W // R), 65535, np.uint16)
        l1, comp = l0.copy(), l0.copy()
        rdf_oracle.eval_forest(d2[None], f0, l0, R, None, None, 1.0)
        rdf_oracle.eval_forest(d2[None], f1, l1, R, l0, 3, 1.0)
        rdf_oracle.composite([l0[0], l1[0]], np.array(conditions, np.int32), comp)
        labels = comp[0][:, ::-1].copy()
        want_means = ms_np.mean_shift(labels[None], 7, variances, 6)
        want_h = ms_np.fingertip_heights(want_means, tips, depth, R, *intr, plane)
        diff_px = int((pipe.labels_image.cu().get() != labels).sum())
        ok = ~np.isnan(want_means)
        dm = float(np.abs(means[ok] - want_means[ok]).max()) if ok.any() else 0.0
        okh = ~np.isnan(want_h)
        same_nan = bool(np.array_equal(np.isnan(means), np.isnan(want_means)) and np.array_equal(np.isnan(heights), np.isnan(want_h)))
        out["parity"] = {"checker": "oracle/ chain of restatements, one hand (flipped) of the frame", "label_pixels": int(labels.size),
                         "differing_label_pixels": diff_px, "labelled_pixels": int((labels != 65535).sum()),
                         "max_abs_mean_diff_px": dm, "nan_pattern_equal": same_nan,
                         "heights_close": bool(np.allclose(heights[okh], want_h[okh], rtol=1e-6, atol=1e-6))}
        assert diff_px == 0 and dm < 1e-9 and same_nan, out["parity"]

    for _ in range(20):
        pipe.run(dbuf, gbuf, 1, False)
    out["us_per_hand_per_frame"] = round(_timed(lambda i: pipe.run(dbuf, gbuf, 1 + (i & 1), bool(i & 1)), n, sync) * 1e6, 1)
    replay = pipe.capture(dbuf, gbuf, 1, False)
    for _ in range(20):
        replay()
    dg = _timed(lambda i: replay(), n, sync)
    out["us_per_hand_per_frame_as_hipgraph"] = round(dg * 1e6, 1)
    out["hands_per_second_as_hipgraph"] = round(1 / dg, 1)
    # both hands of a frame in flight together: two pipelines, two streams, two graphs
    pipe2 = pl.HandPipeline(lf, *args)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    with torch.cuda.stream(s1):
        r1 = pipe.capture(dbuf, gbuf, 1, False)
    with torch.cuda.stream(s2):
        r2 = pipe2.capture(dbuf, gbuf, 2, True)
    sync()

    def both(_=0):
        with torch.cuda.stream(s1):
            r1(read=False)
        with torch.cuda.stream(s2):
            r2(read=False)
        with torch.cuda.stream(s1):
            a = r1.read()
        with torch.cuda.stream(s2):
            b = r2.read()
        return a, b
    seq = (pipe.run(dbuf, gbuf, 1, False), pipe2.run(dbuf, gbuf, 2, True))
    for _ in range(10):      # in flight together they give what they give one after the other
        for (gm, gh), (wm, wh) in zip(both(), seq):
            assert np.array_equal(gm.view(np.uint64), wm.view(np.uint64)) and np.array_equal(gh.view(np.uint64), wh.view(np.uint64))
    out["us_per_frame_both_hands_two_streams"] = round(_timed(both, n, sync) * 1e6, 1)
    # the reference's sequence of separate kernels around the forest (fused_io=False), as a graph, for comparison
    pipe_ref = pl.HandPipeline(lf, *args, fused_io=False)
    r_ref = pipe_ref.capture(dbuf, gbuf, 1, False)
    a, b = r_ref(), replay()
    assert np.array_equal(a[0].view(np.uint64), b[0].view(np.uint64)) and np.array_equal(a[1].view(np.uint64), b[1].view(np.uint64))
    for _ in range(20):
        r_ref()
    out["us_per_hand_per_frame_as_hipgraph_unfused_io"] = round(_timed(lambda i: r_ref(), n, sync) * 1e6, 1)
    return out


def mean_shift(rdf, n=200):
    """rdf_mean_shift on the app's label-map sizes, against the numpy restatement."""
    import torch
    from oracle import mean_shift_numpy as ms_np
    from test_mean_shift import _label_map
    msmod = importlib.import_module("3d-beats_amd.cuda.mean_shift")
    out = {}
    for (h, w) in [(240, 424), (480, 848)]:
        L, rounds = 6, 6       # 3d_bz.py:65, 108-113
        lab = _label_map(11, h, w, L, absent=())
        var = np.full(L, 10.0, np.float32)
        dl, dv = rdf.to_device(lab[None]), rdf.to_device(var)
        ms = msmod.MeanShift()
        for _ in range(5):
            ms.run_device(rounds, dl, L, dv)
        dev = _timed(lambda i: ms.run_device(rounds, dl, L, dv), n, torch.cuda.synchronize)
        host = _timed(lambda i: ms.run(rounds, dl, L, dv), 50, torch.cuda.synchronize)   # + the D2H of the means the reference API returns
        t0 = time.perf_counter()
        want = ms_np.mean_shift(lab, L, var, rounds)
        cpu = time.perf_counter() - t0
        got = ms.run(rounds, dl, L, dv)
        diff = float(np.nanmax(np.abs(got - want)))
        assert diff < 1e-9 and np.array_equal(np.isnan(got), np.isnan(want)), diff
        out[f"{w}x{h}"] = {"device_us_per_run": round(dev * 1e6, 1), "with_result_copy_us": round(host * 1e6, 1),
                           "numpy_restatement_ms": round(cpu * 1e3, 2), "max_abs_diff_px": diff,
                           "rounds": rounds, "classes": L, "launches": 1, "label_bytes_per_round": h * w * 2}
    return out


def train(rdf, images=64, depth=12, proposals=256, blocks=1, noisy_labels=False, check=True):
    """DecisionTreeTrainer.train on synthetic labelled frames.  Unit: (labelled pixel, proposal) feature
    evaluations per second, the work of the histogram kernel.  The check trains a small tree (6 frames, depth 6, 2 x 16
    proposals) on the device and with the numpy restatement from the same proposals and compares them bit for bit."""
    import torch
    from oracle import train_numpy as tn
    from test_training import _ArrayDataset
    h, w, C = 480, 848, 4
    frames = rdf.synth.frames(["live"] * images, 7000, h, w)
    yy, xx = np.mgrid[0:h, 0:w]
    labels = np.zeros(frames.shape, np.uint16)
    for i in range(images):
        valid = (frames[i] != 0) & (frames[i] != 65535)
        cls = 1 + ((xx > w // 2).astype(int) + 2 * (frames[i] > 4000).astype(int)) % 3
        if noisy_labels:
            cls = np.random.default_rng(i).integers(1, C, size=(h, w))
        labels[i][valid] = cls[valid]
    n_lab = int((labels > 0).sum())
    ds = _ArrayDataset(frames, labels, C, per_block=images)
    trainer = rdf.DecisionTreeTrainer(images, proposals)
    trainer.allocate(ds, proposals * blocks, depth)
    tree = rdf.DecisionTree(depth, C)
    np.random.seed(1)
    trainer.train(ds, tree)          # warm-up
    torch.cuda.synchronize()
    times = []
    for _ in range(3):
        np.random.seed(1)
        t0 = time.perf_counter()
        trainer.train(ds, tree)
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    dt = float(np.median(times))
    t = tree.tree_out_cu.get()
    levels = int(np.ceil(np.log2(np.nonzero(np.abs(t).sum(1) > 0)[0].max() + 2)))
    evals = n_lab * proposals * blocks * levels     # upper bound: pixels retire as their nodes become leaves
    per_level = None
    if os.environ.get("RDF_TRAIN_LEVELS"):       # where the time goes, level by level (one synchronisation per level)
        trainer.time_levels = True
        np.random.seed(1)
        trainer.train(ds, tree)
        per_level = [{"level": l, "active_nodes": n, "ms": round(t * 1e3, 2)} for l, n, t in trainer.level_seconds]
        trainer.time_levels = False
    out = {"images": images, "frame": [h, w], "labelled_pixels": n_lab, "classes": C, "max_depth": depth,
           "levels_trained": levels, "proposals_per_level": proposals * blocks, "seconds": round(dt, 4),
           "seconds_of_3_runs": [round(x, 4) for x in times],
           "G_pixel_proposals_per_s_upper_bound": round(evals / dt / 1e9, 2)}
    if per_level is not None:
        out["per_level"] = per_level
    if check:
        D, blk, P, sub = 6, 2, 16, 3
        ds2 = _ArrayDataset(frames[:sub], labels[:sub], C, per_block=3)
        tr2 = rdf.DecisionTreeTrainer(3, P)
        tr2.allocate(ds2, blk * P, D)
        tree2 = rdf.DecisionTree(D, C)
        np.random.seed(7)
        tr2.train(ds2, tree2)
        got = tree2.tree_out_cu.get()

        def props(n):
            arr = np.zeros((n, 5), np.float32)
            importlib.import_module("3d-beats_amd.decision_tree").make_random_features(n, arr)
            return arr
        np.random.seed(7)
        t0 = time.perf_counter()
        want = tn.train_tree(frames[:sub], labels[:sub], C, D, blk, P, proposal_fn=props)
        cpu = time.perf_counter() - t0
        differing = int((got.view(np.uint32) != want.view(np.uint32)).sum())
        out["parity"] = {"checker": "oracle/train_numpy.py, same proposals", "frames": sub, "max_depth": D,
                         "proposals_per_level": blk * P, "tree_words": int(got.size), "differing_words": differing,
                         "numpy_restatement_s": round(cpu, 2)}
        assert differing == 0, out["parity"]
    return out


def trainer_forest(rdf, frames_np, trees=4, depth=20, images=64, proposals=512, train_frames=None, return_forest=False):
    """A forest produced by THIS repo's trainer (DecisionTreeTrainer) instead of a synthetic topology: `trees` trees of depth
    `depth` trained on `images` frames of the bench's mix labelled by a teacher (a balanced tree's labels: something a deep
    tree can fit), evaluated on the headline's batch `frames_np`: nodes the trainer wrote and the batch visits per level, the
    table DecisionForest.tune picks, the batch rate, four frames against the oracle.  (tools/trained_forest_probe.py is the
    stand-alone version.)"""
    import torch
    from oracle import rdf_oracle
    from test_training import _ArrayDataset
    synth = rdf.synth
    n, h, w = frames_np.shape
    C = 4
    t0 = time.perf_counter()
    if train_frames is None:
        train_frames = synth.mixed_batch(images, 20000, h, w)
    images = int(train_frames.shape[0])
    teacher = synth.forest(1, min(16, depth), C, "balanced")
    lab = np.full(train_frames.shape, 65535, np.uint16)
    rdf_oracle.eval_forest(train_frames, teacher, lab)
    labels = np.where(lab == 65535, 0, 1 + lab % (C - 1)).astype(np.uint16)          # 0 = unlabelled, classes 1 .. C-1
    ds = _ArrayDataset(train_frames, labels, C, per_block=images)
    trainer = rdf.DecisionTreeTrainer(images, proposals)
    trainer.allocate(ds, proposals, depth)
    tree = rdf.DecisionTree(depth, C)
    forest_np = np.zeros((trees, (1 << depth) - 1, 7 + 2 * C), np.float32)
    train_s = []
    for k in range(trees):
        np.random.seed(1000 + k)
        tt = time.perf_counter()
        trainer.train(ds, tree)
        torch.cuda.synchronize()
        train_s.append(round(time.perf_counter() - tt, 2))
        forest_np[k] = tree.tree_out_cu.get()
    used = np.abs(forest_np).sum(axis=2) > 0
    written = [int(used[:, (1 << j) - 1:(1 << (j + 1)) - 1].sum()) for j in range(depth)]
    del trainer, ds
    f = rdf.DecisionForest.from_numpy(forest_np)
    depth_dev = rdf.to_device(frames_np)
    tune = f.tune(depth_dev[0:min(32, n)])
    ev = rdf.DecisionTreeEvaluator()
    out = rdf.DeviceArray(frames_np.shape, np.uint16).fill(65535)
    rt = rdf.get_runtime()
    for _ in range(2):
        ev.get_labels_forest(f, depth_dev, out)
    torch.cuda.synchronize()
    ts = []
    for _ in range(7):
        tt = time.perf_counter()
        ev.get_labels_forest(f, depth_dev, out)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - tt)
    ms = float(np.median(ts)) * 1e3
    ns = min(4, n)
    want = np.full((ns, h, w), 65535, np.uint16)
    rdf_oracle.eval_forest(frames_np[0:ns], forest_np, want)
    mism = int((want != out[0:ns].get()).sum())
    assert mism == 0, f"trainer-produced forest: GPU labels differ from the oracle in {mism} pixels"
    visited = rdf_oracle.distinct_nodes_per_level(frames_np[0:min(8, n)], forest_np).sum(axis=0)
    lv = rdf_oracle.walk_lengths(frames_np[0:min(8, n)], forest_np)
    valid = lv.max(axis=3) > 0
    res = {"value": round(n * h * w / ms / 1e3, 2), "unit": "Mpix/s", "ms_per_step": round(ms, 4), "frames": n,
            "forest": f"T{trees}/D{depth}/C{C} trained by DecisionTreeTrainer on {images} teacher-labelled frames, {proposals} proposals per level",
            "train_seconds_per_tree": train_s, "nodes_written_per_level": written,
            "nodes_visited_per_level_by_8_frames": [int(v) for v in visited],
            "share_of_level_visited": [round(float(v) / (trees << j), 4) for j, v in enumerate(visited)],
            "levels_per_pixel_and_tree": round(float(lv[valid].mean()), 2), "tune": tune,
            "parity": {"frames_checked": ns, "differing_pixels": mism, "checker": "oracle/rdf_oracle.c on the host"},
            "setup_seconds": round(time.perf_counter() - t0, 1)}
    return (res, forest_np) if return_forest else res
