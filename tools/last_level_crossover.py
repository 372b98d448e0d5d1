#!/usr/bin/env python3
"""From which share of the deepest level in use does the last-level table pay?  Forests of the bench shape (4 trees, depth 20,
4 classes) whose sides turn into leaves with probability p per level (p = 0: the full topology), evaluated on the bench
batch with the table forced on and off (rdf_set_last_level_table 1 / 0).
usage: tools/last_level_crossover.py [--frames 128] [--probs 0,0.01,...]"""
import argparse
import os
import sys
from importlib import import_module

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=128)
    ap.add_argument("--probs", default="0,0.01,0.02,0.03,0.05,0.08,0.15")
    ap.add_argument("--rounds", type=int, default=7)
    args = ap.parse_args()
    rdf = import_module("3d-beats_amd")
    synth = rdf.synth
    import torch
    rt = rdf.device.get_runtime()
    lib = rt.lib
    T, D, C = 4, 20, 4
    frames = synth.frames(["dense", "live"] * (args.frames // 2), 0)
    depth = rdf.to_device(frames)
    labels = rdf.DeviceArray(frames.shape, np.uint16)
    ev = rdf.DecisionTreeEvaluator(use_packed=True)
    print(f"{args.frames} frames 848x480, T{T} D{D} C{C}; leaf probability per side and level; median of {args.rounds} launches")
    for p in [float(x) for x in args.probs.split(",")]:
        f_np = np.stack([synth.full_tree(k, D, C) if p == 0 else synth.trained_like_tree(k, D, C, leaf_prob=p) for k in range(T)])
        forest = rdf.DecisionForest.from_numpy(f_np)
        packed = forest.packed(1.0)
        rt.synchronize()
        trailer = (T << D) * 80 + (T << (D - 1)) * 64
        unusable, in_use = (int(x) for x in packed.get()[trailer:trailer + 8].view(np.uint32))
        out = {}
        ref = None
        for knob in (1, 0):
            lib.rdf_set_last_level_table(knob)
            ts = []
            for i in range(args.rounds + 2):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                ev.get_labels_forest(forest, depth, labels)
                e1.record()
                e1.synchronize()
                ts.append(e0.elapsed_time(e1))
            out[knob] = float(np.median(ts[2:]))
            got = labels.get()
            if ref is None:
                ref = got
            assert np.array_equal(ref, got)
        lib.rdf_set_last_level_table(-1)
        share = in_use / (T << (D - 1))
        print(f"p {p:5.2f}: level D-1 in use {share:7.4f} (unusable {unusable})   table on {out[1]:7.3f} ms   off {out[0]:7.3f} ms   on/off {out[1] / out[0]:6.3f}")


if __name__ == "__main__":
    main()
