#!/usr/bin/env python3
"""Times DecisionTreeTrainer.train on synthetic labelled frames and the numpy restatement on a
subset.  Unit: (labelled pixel, proposal) feature evaluations per second, the work of the histogram kernel."""
import argparse
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
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=64)
    ap.add_argument("--depth", type=int, default=12)
    ap.add_argument("--proposals", type=int, default=256)
    ap.add_argument("--blocks", type=int, default=1)
    ap.add_argument("--noisy-labels", action="store_true", help="per-pixel random classes (no spatial coherence)")
    a = ap.parse_args()
    import torch
    from oracle import train_numpy as tn
    from test_training import _ArrayDataset
    rdf = importlib.import_module("3d-beats_amd")
    h, w, C = 480, 848, 4
    depth = rdf.synth.frames(["live"] * a.images, 7000, h, w)
    yy, xx = np.mgrid[0:h, 0:w]
    labels = np.zeros(depth.shape, np.uint16)
    for i in range(a.images):
        valid = (depth[i] != 0) & (depth[i] != 65535)
        cls = 1 + ((xx > w // 2).astype(int) + 2 * (depth[i] > 4000).astype(int)) % 3
        if a.noisy_labels:
            cls = np.random.default_rng(i).integers(1, C, size=(h, w))
        labels[i][valid] = cls[valid]
    n_lab = int((labels > 0).sum())
    ds = _ArrayDataset(depth, labels, C, per_block=a.images)
    trainer = rdf.DecisionTreeTrainer(a.images, a.proposals)
    trainer.allocate(ds, a.proposals * a.blocks, a.depth)
    tree = rdf.DecisionTree(a.depth, C)
    np.random.seed(1)
    trainer.train(ds, tree)          # warm-up
    torch.cuda.synchronize()
    np.random.seed(1)
    t0 = time.perf_counter()
    trainer.train(ds, tree)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    t = tree.tree_out_cu.get()
    levels = int(np.ceil(np.log2(np.nonzero(np.abs(t).sum(1) > 0)[0].max() + 2)))
    evals = n_lab * a.proposals * a.blocks * levels     # upper bound: pixels retire as their nodes become leaves
    out = {"train": {"images": a.images, "frame": [h, w], "labelled_pixels": n_lab, "classes": C, "max_depth": a.depth,
                     "levels_trained": levels, "proposals_per_level": a.proposals * a.blocks,
                     "seconds": round(dt, 4), "G_pixel_proposals_per_s_upper_bound": round(evals / dt / 1e9, 2)}}
    # numpy restatement on 2 images, same settings
    sub = 2
    np.random.seed(1)
    t0 = time.perf_counter()
    tn.train_tree(depth[:sub], labels[:sub], C, min(a.depth, 6), 1, 16)
    out["train"]["numpy_restatement_s_for_2_images_D6_16_proposals"] = round(time.perf_counter() - t0, 2)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
