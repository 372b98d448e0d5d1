#!/usr/bin/env python3
"""What one 848x480 frame costs in pieces (round 6): whole, the rows that are exactly ONE round of workgroups (384 with narrow
tiles), the rows a wave-per-tree launch would take, and the two launched back to back -- with narrow tiles on and off.
profiles/r06_latency_pieces.txt"""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
rdf = importlib.import_module("3d-beats_amd")
lib = rdf.get_runtime().lib
forest = rdf.DecisionForest.from_numpy(rdf.synth.forest(4, 20, 4, "full"))
forest.packed(1.0)
ev = rdf.DecisionTreeEvaluator(); ev.auto_tune = False
def t(fn, n=300):
    for _ in range(20): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
for kind in ("dense", "live"):
    frame = rdf.synth.frames([kind], 0, 480, 848)
    d = rdf.to_device(frame)
    for fold in (0, -1):
        lib.rdf_set_fold(fold)
        for rows in (480, 384, 400, 416, 96, 80, 64):
            for tw in ((-1, 0) if rows <= 128 else (-1,)):
                lib.rdf_set_tree_waves(tw)
                dd = rdf.to_device(frame[:, :rows].copy())
                out = rdf.DeviceArray((1, rows, 848), np.uint16).fill(65535)
                us = t(lambda: ev.get_labels_forest(forest, dd, out))
                print(f"{kind} fold {fold:2d} rows {rows:3d} tree_waves {tw:2d}: {us:7.1f} us", flush=True)
        lib.rdf_set_tree_waves(-1)
        # two launches back to back on one stream: top rows four trees per lane, bottom rows a wave per tree
        for top in (384, 400):
            dt, db = rdf.to_device(frame[:, :top].copy()), rdf.to_device(frame[:, top:].copy())
            ot, ob = rdf.DeviceArray((1, top, 848), np.uint16).fill(65535), rdf.DeviceArray((1, 480 - top, 848), np.uint16).fill(65535)
            def both():
                ev.get_labels_forest(forest, dt, ot); ev.get_labels_forest(forest, db, ob)
            print(f"{kind} fold {fold:2d} two launches {top}+{480 - top}: {t(both):7.1f} us", flush=True)
lib.rdf_set_fold(-1)
