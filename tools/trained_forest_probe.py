#!/usr/bin/env python3
"""A forest produced by THIS repo's trainer (DecisionTreeTrainer, csrc/tree_train_hip.hip) instead of a synthetic topology:
is a trained forest's deep working set like synth's "balanced" one (occupied) or like the "full" one (cache-resident)?

Labels come from a teacher -- a balanced forest's own labels on the training frames -- so that the student has something a
deep tree can fit; T trees of depth D are trained on the bench's frame mix with the reference's proposal distribution, the
forest is evaluated on OTHER frames: nodes visited per level (oracle), time with the heap-order table and with the deep
blocks (DecisionForest.tune), labels against the oracle.

    python3 tools/trained_forest_probe.py [--trees 4] [--depth 20] [--images 128] [--proposals 512]
"""
import argparse
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trees", type=int, default=4)
    ap.add_argument("--depth", type=int, default=20)
    ap.add_argument("--images", type=int, default=128)
    ap.add_argument("--proposals", type=int, default=512)
    ap.add_argument("--eval-frames", type=int, default=64)
    a = ap.parse_args()
    import torch
    rdf = importlib.import_module("3d-beats_amd")
    from oracle import rdf_oracle
    from test_training import _ArrayDataset
    synth = rdf.synth
    ev = rdf.DecisionTreeEvaluator()
    h, w, C = 480, 848, 4
    train_frames = synth.mixed_batch(a.images, 20000, h, w)
    teacher = synth.forest(1, 16, C, "balanced")
    lab = np.full(train_frames.shape, 65535, np.uint16)
    rdf_oracle.eval_forest(train_frames, teacher, lab)
    labels = np.where(lab == 65535, 0, lab + 1).astype(np.uint16)        # 0 = unlabelled, classes 1..C... (trainer: ids 0..C-1 with 0 = none)
    labels = np.where(labels > 0, 1 + (labels - 1) % (C - 1), 0).astype(np.uint16)
    ds = _ArrayDataset(train_frames, labels, C, per_block=a.images)
    trainer = rdf.DecisionTreeTrainer(a.images, a.proposals)
    trainer.allocate(ds, a.proposals, a.depth)
    tree = rdf.DecisionTree(a.depth, C)
    forest_np = np.zeros((a.trees, (1 << a.depth) - 1, 7 + 2 * C), np.float32)
    for k in range(a.trees):
        np.random.seed(1000 + k)
        t0 = time.perf_counter()
        trainer.train(ds, tree)
        torch.cuda.synchronize()
        forest_np[k] = tree.tree_out_cu.get()
        used = np.abs(forest_np[k]).sum(1) > 0
        per_level = [int(used[(1 << j) - 1:(1 << (j + 1)) - 1].sum()) for j in range(a.depth)]
        print(f"tree {k}: trained in {time.perf_counter() - t0:.2f} s; nodes written per level: {per_level}", flush=True)
    frames = synth.mixed_batch(a.eval_frames, 0, h, w)
    dn = rdf_oracle.distinct_nodes_per_level(frames[:8], forest_np).sum(axis=0)
    lv = rdf_oracle.walk_lengths(frames[:8], forest_np)
    valid = lv.max(axis=3) > 0
    print("nodes visited per level by 8 evaluation frames (all trees):", [int(v) for v in dn])
    print(f"levels per (pixel, tree): {lv[valid].mean():.2f}; share of walks that reach level D-1: {(lv[valid] == a.depth).mean():.3f}")
    f = rdf.DecisionForest.from_numpy(forest_np)
    depth = rdf.to_device(frames)
    res = f.tune(depth[0:32])
    print("tune:", res)
    out = rdf.DeviceArray(frames.shape, np.uint16).fill(65535)
    lib = rdf.get_runtime().lib
    for name, level in (("heap-order", 0), ("tuned", res["deep_from"])):
        lib.rdf_forest_set_deep_from(f.packed(1.0).ptr, level)
        for _ in range(2):
            ev.get_labels_forest(f, depth, out)
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            ev.get_labels_forest(f, depth, out)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        print(f"{name:10s} (deep_from {level:2d}): {np.median(ts) * 1e3:7.3f} ms for {a.eval_frames} frames = "
              f"{a.eval_frames * h * w / np.median(ts) / 1e6:8.1f} Mpix/s", flush=True)
    want = np.full((4, h, w), 65535, np.uint16)
    rdf_oracle.eval_forest(frames[:4], forest_np, want)
    print("labels of 4 frames against the oracle:", int((want != out[0:4].get()).sum()), "differing pixels")


if __name__ == "__main__":
    main()
