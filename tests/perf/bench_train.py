#!/usr/bin/env python3
"""Times DecisionTreeTrainer.train on synthetic labelled frames.  Command line of tools/bench_legs.train (bench.py runs the same function as its `train` leg)."""
import argparse
import importlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=64)
    ap.add_argument("--depth", type=int, default=12)
    ap.add_argument("--proposals", type=int, default=256)
    ap.add_argument("--blocks", type=int, default=1)
    ap.add_argument("--noisy-labels", action="store_true", help="per-pixel random classes (no spatial coherence)")
    a = ap.parse_args()
    import bench_legs
    rdf = importlib.import_module("3d-beats_amd")
    print(json.dumps({"train": bench_legs.train(rdf, a.images, a.depth, a.proposals, a.blocks, a.noisy_labels)}))


if __name__ == "__main__":
    main()
