#!/usr/bin/env python3
"""The app's real shapes beside the bench's: classes 3 / 4 / 7 / 8, three and four trees, labels_reduce 1 and 2
(/root/reference/src/3d_bz.py:49, 76, 108-110 runs T3-4, 7 composite classes, r = 2), on the "full" (cache-resident) and
the "balanced" (occupied) topology, every forest tuned on the batch's first frames (DecisionForest.tune).  Prints ms per
launch, Mpix/s and nanoseconds per node visit relative to the four-class forest of the same trees / reduce / topology -- the
no-cliff table of VERDICT r3 item 3.

    python3 tools/class_sweep.py [--frames 64] [--depth 18]
"""
import argparse
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=64)
    ap.add_argument("--depth", type=int, default=18)
    ap.add_argument("--rounds", type=int, default=5)
    a = ap.parse_args()
    import torch
    rdf = importlib.import_module("3d-beats_amd")
    rdf.get_runtime()
    ev = rdf.DecisionTreeEvaluator()
    host = rdf.synth.mixed_batch(a.frames, 0, 480, 848)
    depth = rdf.to_device(host)
    valid = int(((host != 0) & (host != 65535)).sum())
    print(f"{a.frames} mixed 848x480 frames, depth {a.depth}; {valid} valid pixels at labels_reduce 1")
    print(f"{'topology':9s} {'T':>2s} {'C':>2s} {'r':>2s} {'tuned':>6s} {'ms':>8s} {'Mpix/s':>9s} {'ns/visit/lane':>14s} {'vs C=4':>7s}")
    for topology in ("full", "balanced"):
        for T in (3, 4):
            for r in (1, 2):
                base = None
                for C in (4, 3, 7, 8):
                    forest = rdf.DecisionForest.from_numpy(rdf.synth.forest(T, a.depth, C, topology))
                    forest.packed(1.0)
                    tune = forest.tune(depth[0:32], labels_reduce=r)
                    labels = rdf.DeviceArray((a.frames, 480 // r, 848 // r), np.uint16).fill(65535)
                    for _ in range(2):
                        ev.get_labels_forest(forest, depth, labels, r)
                    torch.cuda.synchronize()
                    ts = []
                    for _ in range(a.rounds):
                        t0 = time.perf_counter()
                        ev.get_labels_forest(forest, depth, labels, r)
                        torch.cuda.synchronize()
                        ts.append(time.perf_counter() - t0)
                    ms = float(np.median(ts)) * 1e3
                    v = int((host[:, ::r, ::r][:, :480 // r, :848 // r] != 0).sum()) if False else None
                    sub = host[:, 0:(480 // r) * r:r, 0:(848 // r) * r:r]
                    visits = int(((sub != 0) & (sub != 65535)).sum()) * T * a.depth
                    ns = ms * 1e6 / visits * 256 * 24 * 64 / 1.0      # ns per visit per resident lane (256 CUs x 24 waves x 64 lanes)
                    base = ns if C == 4 else base
                    print(f"{topology:9s} {T:2d} {C:2d} {r:2d} {tune['deep_from']:6d} {ms:8.3f} {a.frames * 480 * 848 / ms / 1e3:9.1f} "
                          f"{ns:14.1f} {ns / base:7.2f}", flush=True)
                    del forest, labels


if __name__ == "__main__":
    main()
