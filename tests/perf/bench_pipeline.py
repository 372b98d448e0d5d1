#!/usr/bin/env python3
"""Per-hand, per-frame latency of the app's chain (3d_bz.py:388-522) through HandPipeline.  Command line of tools/bench_legs.hand_pipeline (bench.py runs the same function as its `hand_pipeline` leg)."""
import importlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    import bench_legs
    rdf = importlib.import_module("3d-beats_amd")
    print(json.dumps({"hand_pipeline": bench_legs.hand_pipeline(rdf)}))


if __name__ == "__main__":
    main()
