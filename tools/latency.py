#!/usr/bin/env python3
"""Single-frame latency of the forest kernel (config 2's launch: ONE 848x480 frame, T4/D20/C4) under the small-launch
knobs: tree waves, rows per wave, block size, halo, LDS levels.  Every variant's labels are
compared with the first one's.   usage: python3 tools/latency.py [--kind dense|live] [--reduce 1] [--topology full]"""
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
    ap.add_argument("--kind", default="dense")
    ap.add_argument("--reduce", type=int, default=1)
    ap.add_argument("--topology", default="full")
    ap.add_argument("--trees", type=int, default=4)
    ap.add_argument("--depth", type=int, default=20)
    ap.add_argument("--n", type=int, default=300)
    ap.add_argument("--deep", type=int, nargs="*", default=None, help="compare the deep-block walks from these root levels instead")
    a = ap.parse_args()
    import torch
    rdf = importlib.import_module("3d-beats_amd")
    lib = rdf.get_runtime().lib
    forest = rdf.DecisionForest.from_numpy(rdf.synth.forest(a.trees, a.depth, 4, a.topology))
    forest.packed(1.0)
    frame = rdf.synth.frames([a.kind], 0, 480, 848)
    depth = rdf.to_device(frame)
    r = a.reduce
    labels = rdf.DeviceArray((1, 480 // r, 848 // r), np.uint16).fill(65535)
    ev = rdf.DecisionTreeEvaluator()
    ev.auto_tune = False
    variants = [("default", {}), ("tree waves off", {"tree_waves": 0}), ("rows per wave 2", {"rows_per_wave": 2}),
                ("512 threads", {"block_threads": 512}), ("halo 16", {"halo": 16}), ("halo 24", {"halo": 24}),
                ("lds levels 6", {"lds_levels": 6}), ("lds levels 8", {"lds_levels": 8}),
                # more, smaller workgroup footprints: every tile of the frame resident at once (one round instead of 1.3)
                ("2 trees/lane, 23 KB, halo 24", {"group": 2, "lds_budget_bytes": 23000, "halo": 24}),
                ("2 trees/lane, 23 KB, halo 24, 6 levels", {"group": 2, "lds_budget_bytes": 23000, "halo": 24, "lds_levels": 6}),
                ("2 trees/lane, 23 KB, halo 16", {"group": 2, "lds_budget_bytes": 23000, "halo": 16}),
                ("2 trees/lane, default LDS", {"group": 2}),
                ("1 tree/lane, 20 KB, halo 16", {"group": 1, "lds_budget_bytes": 20000, "halo": 16}),
                ("1 tree/lane, 20 KB, halo 24, 6 levels", {"group": 1, "lds_budget_bytes": 20000, "halo": 24, "lds_levels": 6}),
                ("4 trees/lane, 26 KB, halo 24", {"lds_budget_bytes": 26000, "halo": 24})]
    if a.deep:      # the deep-block walk of a small launch, by take-over level and geometry
        variants = [("heap-order records", {"deep_from": 0})]
        for lvl in a.deep:
            variants += [(f"deep {lvl}", {"deep_from": lvl}), (f"deep {lvl} 512 threads", {"deep_from": lvl, "block_threads": 512}),
                         (f"deep {lvl} halo 16", {"deep_from": lvl, "halo": 16, "lds_budget_bytes": 21000}),
                         (f"deep {lvl} rows 2", {"deep_from": lvl, "rows_per_wave": 2})]
    defaults = {"tree_waves": -1, "rows_per_wave": 0, "block_threads": 0, "halo": -1, "lds_levels": -1, "deep_from": -1, "lds_budget_bytes": 0, "group": 0}
    ref = None
    for name, knobs in variants:
        for k, v in {**defaults, **knobs}.items():
            getattr(lib, "rdf_set_" + k)(v)
        labels.fill(65535)
        for _ in range(20):
            ev.get_labels_forest(forest, depth, labels, r)
        torch.cuda.synchronize()
        ts = []
        for rep in range(5):
            t0 = time.perf_counter()
            for _ in range(a.n):
                ev.get_labels_forest(forest, depth, labels, r)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) / a.n)
        got = labels.get()
        ref = got if ref is None else ref
        same = bool(np.array_equal(got, ref))
        print(f"{a.kind} r={r} T{a.trees}/D{a.depth} {a.topology:8s} {name:34s} median {np.median(ts) * 1e6:7.1f} us  min {min(ts) * 1e6:7.1f} us  "
              f"labels {'equal' if same else 'DIFFER'}", flush=True)
        assert same
    for k, v in defaults.items():
        getattr(lib, "rdf_set_" + k)(v)


if __name__ == "__main__":
    main()
