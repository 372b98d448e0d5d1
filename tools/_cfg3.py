#!/usr/bin/env python3
"""scratch: config 3's frame (2-layer run, r=2, live frame) and one r=2 forest launch, wall per call"""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
rdf = importlib.import_module("3d-beats_amd")
forest = rdf.DecisionForest.from_numpy(rdf.synth.forest(4, 20, 4, "full"))
forest.packed(1.0)
H, W = 480, 848
frames = rdf.synth.mixed_batch(2, 0, H, W)
cfg3 = {"layers": [{"model": forest}, {"model": forest, "filter_model": 0, "filter_model_class": 3}],
        "conditions": [[0, 1], [0, 2], [1, 3], [0, 3], [0, 4], [0, 5], [0, 6]], "label_colors": [[0, 0, 0, 255]] * 6}
lf = rdf.LayeredDecisionForest(cfg3, (H, W), 2)
dbuf, lbuf = rdf.GpuBuffer((H, W), np.uint16), rdf.GpuBuffer((H // 2, W // 2), np.uint16)
dbuf.cu().set(frames[1])
ev = rdf.DecisionTreeEvaluator(); ev.auto_tune = False
d1 = rdf.to_device(frames[1:2]); o1 = rdf.DeviceArray((1, H // 2, W // 2), np.uint16).fill(65535)
def t(fn, n=400):
    for _ in range(30): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
for rep in range(3):
    print(f"cfg3 layered run {t(lambda: lf.run(dbuf, lbuf, 1.0)):6.1f} us   one forest launch r=2 live {t(lambda: ev.get_labels_forest(forest, d1, o1, 2)):6.1f} us", flush=True)
