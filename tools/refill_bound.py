#!/usr/bin/env python3
"""What refilling lanes could save AT MOST on a forest whose walks end early -- from the batch's own walk
lengths (oracle/rdf_oracle.c), not from an independence argument.  The forest kernel's wave = 64 consecutive label pixels of
a row; it runs the level body until its longest walk has ended.  With lanes refilled the moment a pixel's four walks have
ended (and at no cost) it would run the MEAN over its lanes of a pixel's longest walk.  CPU only.

    python3 tools/refill_bound.py [--frames 16] [--topology trained]
"""
import argparse
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=16)
    ap.add_argument("--topology", default="trained")
    ap.add_argument("--trees", type=int, default=4)
    ap.add_argument("--depth", type=int, default=20)
    a = ap.parse_args()
    synth = importlib.import_module("3d-beats_amd.synth")
    from oracle import rdf_oracle
    forest = synth.forest(a.trees, a.depth, 4, a.topology)
    frames = synth.mixed_batch(a.frames, 0, 480, 848)
    lv = rdf_oracle.walk_lengths(frames, forest)                      # [N, H, W, T]
    n, h, w, T = lv.shape
    longest = lv.max(axis=3).astype(np.float64)                       # a pixel's longest walk = its level-loop iterations
    valid = longest > 0
    pad = (-w) % 64
    lw = np.pad(longest, ((0, 0), (0, 0), (0, pad))).reshape(n, h, -1, 64)
    vw = np.pad(valid, ((0, 0), (0, 0), (0, pad))).reshape(n, h, -1, 64)
    wave_max = lw.max(axis=3)                                          # what a wave runs today
    live = vw.any(axis=3)
    lanes = vw.sum(axis=3)
    wave_mean_valid = np.where(live, lw.sum(axis=3) / np.maximum(lanes, 1), 0.0)       # refill with the pixels compacted too
    wave_mean_64 = lw.sum(axis=3) / 64.0                                               # refill, idle lanes stay idle
    # two rows per lane (the throughput shape: a wave owns two rows of its tile): a lane can be refilled once
    two = lw[:, : (h // 2) * 2].reshape(n, h // 2, 2, -1, 64)
    both = (two.max(axis=4) > 0).all(axis=2)                            # wave slots whose two rows both hold pixels
    sum_of_max = two.max(axis=4).sum(axis=2)[both]
    max_of_sum = two.sum(axis=2).max(axis=3)[both]
    print(f"{a.topology} T{a.trees}/D{a.depth}, {a.frames} mixed 848x480 frames: {int(valid.sum())} pixels evaluated, "
          f"{lv[valid].mean():.2f} levels per (pixel, tree), {longest[valid].mean():.2f} levels in a pixel's longest walk")
    print(f"level-loop iterations per live wave today (its longest walk):        {wave_max[live].mean():6.2f}")
    print(f"  with free refill, idle lanes staying idle (mean over 64 lanes):     {wave_mean_64[live].mean():6.2f}")
    print(f"  with free refill AND compaction (mean over the valid lanes):        {wave_mean_valid[live].mean():6.2f}"
          f"   -> at most {wave_max[live].sum() / wave_mean_valid[live].sum():.2f} x")
    print(f"two rows per lane, refilled once: max of sums {max_of_sum.mean():6.2f} against sum of maxima {sum_of_max.mean():6.2f}"
          f"   -> at most {sum_of_max.sum() / max_of_sum.sum():.3f} x")


if __name__ == "__main__":
    main()
