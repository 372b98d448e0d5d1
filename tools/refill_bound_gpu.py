#!/usr/bin/env python3
"""VERDICT r3 item 4, on the GPU: what would refilling lanes buy on a forest whose walks end early, AT MOST?

A timing-only experiment on a build with -DRDF_EXPERIMENT_REFILL_BOUND (tools/bin/librdf_refill_bound.so; not the product
build): pass 1 records, for every wave slot (an image row x a 64-column chunk), the sum and the count of its lanes' longest
walks; pass 2 runs every wave's level loop only for the MEAN of them, which is what a wave would run if a lane took the next
pixel the moment its four walks have ended, at no cost and from an unbounded supply (labels are wrong in pass 2: walks are cut
short).  The ratio of the two timings is the zero-overhead bound of lane refill, measured on the kernel itself.

The experiment's code is not in the product source: it is kept as profiles/r04_refill_bound.patch (round 5 moved it out).


    git show 585fd4a:3d-beats_amd/csrc/rdf_hip.hip > (scratch tree)/3d-beats_amd/csrc/rdf_hip.hip
    hipcc ... -DRDF_EXPERIMENT_REFILL_BOUND -o tools/bin/librdf_refill_bound.so <the four .hip files of the scratch tree>
    python3 tools/refill_bound_gpu.py [--topology trained] [--frames 128]
"""
import argparse
import ctypes
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("RDF_HIP_LIBRARY", os.path.join(ROOT, "tools", "bin", "librdf_refill_bound.so"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--topology", default="trained")
    ap.add_argument("--frames", type=int, default=128)
    ap.add_argument("--depth", type=int, default=20)
    a = ap.parse_args()
    import torch
    rdf = importlib.import_module("3d-beats_amd")
    lib = rdf.get_runtime().lib
    lib.rdf_experiment_refill_bound.argtypes = [ctypes.c_void_p] * 3
    lib.rdf_experiment_refill_bound.restype = None
    h, w, T = 480, 848, 4
    forest = rdf.DecisionForest.from_numpy(rdf.synth.forest(T, a.depth, 4, a.topology))
    forest.packed(1.0)
    lib.rdf_forest_set_deep_from(forest.packed(1.0).ptr, 0)
    frames = rdf.synth.mixed_batch(a.frames, 0, h, w)
    depth = rdf.to_device(frames)
    labels = rdf.DeviceArray(frames.shape, np.uint16).fill(65535)
    ev = rdf.DecisionTreeEvaluator()
    tiles_x = (w + 63) // 64
    n_slots = a.frames * h * tiles_x

    def timed(n=7):
        for _ in range(2):
            ev.get_labels_forest(forest, depth, labels)
        torch.cuda.synchronize()
        ts = []
        for _ in range(n):
            t0 = time.perf_counter()
            ev.get_labels_forest(forest, depth, labels)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        return float(np.median(ts)) * 1e3

    lib.rdf_experiment_refill_bound(None, None, None)
    t_now = timed()
    ref = labels.get()
    s = torch.zeros(n_slots, dtype=torch.int32, device="cuda")
    c = torch.zeros(n_slots, dtype=torch.int32, device="cuda")
    lib.rdf_experiment_refill_bound(s.data_ptr(), c.data_ptr(), None)
    ev.get_labels_forest(forest, depth, labels)
    torch.cuda.synchronize()
    lib.rdf_experiment_refill_bound(None, None, None)
    assert np.array_equal(labels.get(), ref)
    s_h, c_h = s.cpu().numpy().astype(np.int64), c.cpu().numpy().astype(np.int64)
    live = c_h > 0
    print(f"{a.topology} T{T}/D{a.depth}, {a.frames} mixed frames: {int(live.sum())} live wave slots, {int(c_h.sum())} pixels; "
          f"mean longest walk per pixel {s_h.sum() / c_h.sum():.2f} levels")
    print(f"kernel as it is:                                             {t_now:7.3f} ms")
    for name, lim in (("free refill + compaction (mean over the valid lanes)", np.ceil(s_h / np.maximum(c_h, 1))),
                      ("free refill, idle lanes stay idle (mean over 64 lanes)", np.ceil(s_h / 64.0))):
        lim8 = torch.from_numpy(np.clip(lim, 0, 255).astype(np.uint8)).cuda()
        lib.rdf_experiment_refill_bound(None, None, lim8.data_ptr())
        t = timed()
        lib.rdf_experiment_refill_bound(None, None, None)
        print(f"{name + ':':61s}{t:7.3f} ms   -> at most {t_now / t:.3f} x   (mean limit {lim[live].mean():.2f} levels; labels differ in "
              f"{int((labels.get() != ref).sum())} pixels, as they must)")


if __name__ == "__main__":
    main()
