#!/usr/bin/env python3
"""Times rdf_mean_shift on the app's label-map sizes and against the numpy restatement.  Command line of tools/bench_legs.mean_shift (bench.py runs the same function as its `mean_shift` leg)."""
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
    print(json.dumps({"mean_shift": bench_legs.mean_shift(rdf)}))


if __name__ == "__main__":
    main()
