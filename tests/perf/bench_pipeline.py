#!/usr/bin/env python3
"""Per-hand, per-frame latency of the app's chain (3d_bz.py:388-522) through HandPipeline: stencil, flip,
0->65535, 2-layer forest (labels_reduce 2), flip back, RGBA, 6 mean-shift rounds, fingertip heights, one copy of
the result to the host.  848x480 frame with two synthetic hands."""
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch
    from test_pipeline import H, W, R, _scene
    rdf = importlib.import_module("3d-beats_amd")
    pl = importlib.import_module("3d-beats_amd.pipeline")
    depth, groups = _scene(rdf)
    # forests of the size the reference's models have (readme: depth 18-20), 7 composite classes
    f0, f1 = rdf.synth.forest(4, 18, 4, "trained", 60), rdf.synth.forest(4, 18, 5, "trained", 70)
    cfg = {"layers": [{"model": rdf.DecisionForest.from_numpy(f0)},
                      {"model": rdf.DecisionForest.from_numpy(f1), "filter_model": 0, "filter_model_class": 3}],
           "conditions": [[0, 1], [0, 2], [1, 3], [0, 3], [0, 4], [0, 5], [0, 6], [0, 7]],
           "label_colors": [[10 * i, 255 - 10 * i, i, 255] for i in range(1, 8)]}
    lf = rdf.LayeredDecisionForest(cfg, (H, W), R)
    pipe = pl.HandPipeline(lf, (H, W), R, 1.0, 6, np.full(7, 40., np.float32), [3, 4, 5, 6, 7],
                           (421.3, 420.9, 423.1, 238.6), np.eye(4, dtype=np.float32))
    dbuf, gbuf = rdf.GpuBuffer((H, W), np.uint16), rdf.GpuBuffer((H, W), np.uint16)
    dbuf.cu().set(depth)
    gbuf.cu().set(groups)
    for _ in range(20):
        pipe.run(dbuf, gbuf, 1, False)
    torch.cuda.synchronize()
    n = 200
    t0 = time.perf_counter()
    for i in range(n):
        pipe.run(dbuf, gbuf, 1 + (i & 1), bool(i & 1))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    replays = [pipe.capture(dbuf, gbuf, 1, False), ]
    for _ in range(20):
        replays[0]()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        replays[0]()
    torch.cuda.synchronize()
    dg = (time.perf_counter() - t0) / n
    # both hands of a frame in flight together: two pipelines, two streams, two graphs
    pipe2 = pl.HandPipeline(lf, (H, W), R, 1.0, 6, np.full(7, 40., np.float32), [3, 4, 5, 6, 7],
                            (421.3, 420.9, 423.1, 238.6), np.eye(4, dtype=np.float32))
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    with torch.cuda.stream(s1):
        r1 = pipe.capture(dbuf, gbuf, 1, False)
    with torch.cuda.stream(s2):
        r2 = pipe2.capture(dbuf, gbuf, 2, True)
    torch.cuda.synchronize()

    def both():
        with torch.cuda.stream(s1):
            r1(read=False)
        with torch.cuda.stream(s2):
            r2(read=False)
        with torch.cuda.stream(s1):
            a = r1.read()
        with torch.cuda.stream(s2):
            b = r2.read()
        return a, b
    # each pipeline owns the label buffers its graph writes (pipe2 took a sibling of `lf`): the two hands in flight together
    # must give what they give one after the other
    seq = (pipe.run(dbuf, gbuf, 1, False), pipe2.run(dbuf, gbuf, 2, True))
    for _ in range(20):
        got = both()
        for (gm, gh), (wm, wh) in zip(got, seq):
            assert np.array_equal(gm.view(np.uint64), wm.view(np.uint64)) and np.array_equal(gh.view(np.uint64), wh.view(np.uint64))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        both()
    torch.cuda.synchronize()
    d2 = (time.perf_counter() - t0) / n
    # the reference's sequence of separate kernels around the forest (fused_io=False), as a graph, for comparison
    pipe_ref = pl.HandPipeline(lf, (H, W), R, 1.0, 6, np.full(7, 40., np.float32), [3, 4, 5, 6, 7],
                               (421.3, 420.9, 423.1, 238.6), np.eye(4, dtype=np.float32), fused_io=False)
    r_ref = pipe_ref.capture(dbuf, gbuf, 1, False)
    a, b = r_ref(), replays[0]()
    assert np.array_equal(a[0].view(np.uint64), b[0].view(np.uint64)) and np.array_equal(a[1].view(np.uint64), b[1].view(np.uint64))
    for _ in range(20):
        r_ref()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        r_ref()
    torch.cuda.synchronize()
    dref = (time.perf_counter() - t0) / n
    print(json.dumps({"hand_pipeline": {"frame": [H, W], "labels_reduce": R, "layers": 2, "trees": 4, "tree_depth": 18,
                                        "us_per_hand_per_frame_as_hipgraph_unfused_io": round(dref * 1e6, 1),
                                        "mean_shift_rounds": 6, "us_per_hand_per_frame": round(dt * 1e6, 1),
                                        "us_per_hand_per_frame_as_hipgraph": round(dg * 1e6, 1),
                                        "hands_per_second_as_hipgraph": round(1 / dg, 1),
                                        "us_per_frame_both_hands_two_streams": round(d2 * 1e6, 1)}}))


if __name__ == "__main__":
    main()
