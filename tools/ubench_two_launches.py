#!/usr/bin/env python3
"""Do two small forest launches (one live 848x480 frame each, labels_reduce 2) overlap when they are issued on two streams,
eagerly and as two branches of one hipGraph?  Measured on MI355X: 101 us for the pair on one stream, 88 us on two streams;
97 and 83 us as graphs -- they mostly serialise, so evaluating the layers of a stack concurrently (unfiltered, filters
applied in the composite kernel; built, bit-exact, then removed) bought 3 us of 81 on config 3 and 2 us of 161 on the
per-hand pipeline."""
import importlib, time, numpy as np, torch, sys
sys.path.insert(0, ".")
rdf = importlib.import_module("3d-beats_amd")
f = rdf.DecisionForest.from_numpy(rdf.synth.forest(4, 20, 4, "full"))
host = rdf.synth.frames(["live"], 1, 480, 848)
depth = rdf.to_device(host)
la, lb = rdf.DeviceArray((1, 240, 424), np.uint16).fill(65535), rdf.DeviceArray((1, 240, 424), np.uint16).fill(65535)
ev = rdf.DecisionTreeEvaluator()
f.packed(1.0)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def seq():
    with torch.cuda.stream(s1):
        ev.get_labels_forest(f, depth, la, 2); ev.get_labels_forest(f, depth, lb, 2)
def par():
    with torch.cuda.stream(s1): ev.get_labels_forest(f, depth, la, 2)
    with torch.cuda.stream(s2): ev.get_labels_forest(f, depth, lb, 2)
for name, fn in (("sequential on one stream", seq), ("two streams", par)):
    for _ in range(20): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(200):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print(f"{name}: median {np.median(ts)*1e6:.1f} us per pair (host-synchronised)")
# as graphs
for name, fn in (("graph sequential", seq), ("graph two streams", None)):
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    with torch.cuda.graph(g, stream=side):
        if fn: 
            ev.get_labels_forest(f, depth, la, 2); ev.get_labels_forest(f, depth, lb, 2)
        else:
            e = torch.cuda.Event(); e.record(side)
            s2.wait_event(e)
            ev.get_labels_forest(f, depth, la, 2)
            with torch.cuda.stream(s2):
                ev.get_labels_forest(f, depth, lb, 2)
                e2 = torch.cuda.Event(); e2.record(s2)
            side.wait_event(e2)
    for _ in range(20): g.replay()
    torch.cuda.synchronize(); ts = []
    for _ in range(200):
        t0 = time.perf_counter(); g.replay(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print(f"{name}: median {np.median(ts)*1e6:.1f} us per pair")
