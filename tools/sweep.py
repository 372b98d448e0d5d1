#!/usr/bin/env python3
"""Interleaved A/B timing of launch geometries (block threads x LDS budget) of the forest kernel on
the bench workload.  usage: python3 tools/sweep.py [--frames 128] [--rounds 5] block:lds ..."""
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
    ap.add_argument("--frames", type=int, default=128)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--reps", type=int, default=1, help="launches per timing sample")
    ap.add_argument("--depth", type=int, default=20)
    ap.add_argument("--trees", type=int, default=4)
    ap.add_argument("--topology", default="full")
    ap.add_argument("--unpacked", action="store_true")
    ap.add_argument("--kinds", default="mixed", choices=["mixed", "interleaved", "dense", "live"])
    ap.add_argument("--reduce", type=int, default=1)
    ap.add_argument("--classes", type=int, default=4)
    ap.add_argument("--filter", type=int, default=0, help="filter on a synthetic earlier layer whose class 1 covers 1/N of "
                    "the frame in 32x32 blocks (0: no filter)")
    ap.add_argument("--no-compaction", action="store_true")
    ap.add_argument("--check", type=int, default=0, help="compare the first N frames' labels with the oracle")
    ap.add_argument("--no-compare", action="store_true", help="timing-only builds: the combos' labels need not agree")
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--width", type=int, default=848)
    ap.add_argument("combos", nargs="*", default=["0:0", "256:32700", "512:54600"], help="0:0 = every default")
    a = ap.parse_args()
    import torch
    rdf = importlib.import_module("3d-beats_amd")
    rt = rdf.get_runtime()
    lib = rt.lib
    forest_np = rdf.synth.forest(a.trees, a.depth, a.classes, a.topology)
    forest = rdf.DecisionForest.from_numpy(forest_np)
    kinds = {"mixed": None, "interleaved": ["dense", "live"] * (a.frames // 2) + ["dense"] * (a.frames % 2),
             "dense": ["dense"] * a.frames, "live": ["live"] * a.frames}[a.kinds]
    host = (rdf.synth.mixed_batch(a.frames, 0, a.height, a.width) if kinds is None
            else rdf.synth.frames(kinds, 0, a.height, a.width))
    depth = rdf.to_device(host)
    red = a.reduce
    labels = rdf.DeviceArray((host.shape[0], host.shape[1] // red, host.shape[2] // red), np.uint16).fill(65535)
    ev = rdf.DecisionTreeEvaluator(use_packed=not a.unpacked)
    ev.auto_tune = False        # (the combos say which table is walked)
    filt = None
    if a.filter:
        yy, xx = np.mgrid[0:host.shape[1] // red, 0:host.shape[2] // red]
        f2d = (((xx // 32 + yy // 32) % a.filter) == 0).astype(np.uint16)
        filt = rdf.to_device(np.broadcast_to(f2d, (host.shape[0],) + f2d.shape).copy())
    lib.rdf_set_compaction(0 if a.no_compaction else -1)
    combos = [tuple(int(x) for x in c.split(":")) for c in a.combos]   # block:lds[:rows_per_wave[:halo[:lds_levels[:stage_vec[:group[:deep_from]]]]]]
    res = {c: [] for c in combos}
    ref = None
    for r in range(a.rounds + 1):
        for c in combos:
            lib.rdf_set_block_threads(c[0])
            lib.rdf_set_lds_budget_bytes(c[1])      # 0 = the library's default for the workgroup size
            lib.rdf_set_rows_per_wave(c[2] if len(c) > 2 else 0)
            lib.rdf_set_halo(c[3] if len(c) > 3 else -1)
            lib.rdf_set_lds_levels(c[4] if len(c) > 4 else -1)
            lib.rdf_set_stage_vec(c[5] if len(c) > 5 else -1)
            lib.rdf_set_group(c[6] if len(c) > 6 else 0)
            lib.rdf_set_deep_from(c[7] if len(c) > 7 else -1)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.reps):
                ev.get_labels_forest(forest, depth, labels, red, filt, 1 if filt is not None else None)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / a.reps
            if r:
                res[c].append(dt * 1e3)
            else:
                got = labels.get()
                ref = got if ref is None else ref
                assert a.no_compare or np.array_equal(got, ref), c
    if a.check:
        from oracle import rdf_oracle
        n = min(a.check, host.shape[0])
        want = np.full((n,) + ref.shape[1:], 65535, np.uint16)
        rdf_oracle.eval_forest(host[:n], forest_np, want, red, None if filt is None else filt.get()[:n], 1 if filt is not None else None)
        print(f"oracle check on {n} frames: {int((want != ref[:n]).sum())} differing pixels", flush=True)
    npx = a.frames * a.height * a.width
    for c in combos:
        v = np.array(res[c])
        print(f"block {c[0]:5d} lds {c[1]:7d} rpw {c[2] if len(c) > 2 else 0} halo {c[3] if len(c) > 3 else -1:3d} levels {c[4] if len(c) > 4 else -1:2d} vec {c[5] if len(c) > 5 else -1:2d} group {c[6] if len(c) > 6 else 0} deep {c[7] if len(c) > 7 else -1:2d}: median {np.median(v):8.3f} ms  min {v.min():8.3f} ms  "
              f"{npx / np.median(v) / 1e3:8.1f} Mpix/s", flush=True)


if __name__ == "__main__":
    main()
