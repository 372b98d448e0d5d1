#!/usr/bin/env python3
"""HostFramesEvaluator under `rocprofv3 --kernel-trace --memory-copy-trace`: ten steps of the bench batch, then (--parse DIR) a
timeline of each step's upload and kernel from the trace: how much of every upload runs under the kernel of the step before.
usage: rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d OUT -- python3 tools/host_stream_trace.py
       python3 tools/host_stream_trace.py --parse OUT"""
import csv
import glob
import os
import sys
from importlib import import_module

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run():
    import torch
    rdf = import_module("3d-beats_amd")
    synth = rdf.synth
    F, H, W = 128, 480, 848
    frames = synth.frames(["dense", "live"] * (F // 2), 0)
    forest = rdf.DecisionForest.from_numpy(synth.forest(4, 20, 4, "full"))
    hp = rdf.HostFramesEvaluator(forest, (F, H, W))
    for b in range(2):
        hp.frames[b][:] = frames
    last = None
    for _ in range(12):
        hp.next_frames()
        last = hp.submit()
    hp.result(last)
    torch.cuda.synchronize()
    print("done: 12 steps")


def parse(d):
    kt = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))
    mt = sorted(glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True))
    assert kt and mt, "no traces found"
    kern = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(kt[-1]))
            if "k_eval_forest" in r["Kernel_Name"]]
    cps = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Direction", r.get("Name", "")))
           for r in csv.DictReader(open(mt[-1]))]
    big = [c for c in cps if c[1] - c[0] > 500_000]          # the 104-MB uploads (the counters' copies are microseconds)
    kern.sort()
    big.sort()
    t0 = kern[0][0]
    print(f"{len(kern)} forest launches, {len(big)} large copies ({len(cps)} copies in all); times in ms from the first launch")
    print("step  kernel start..end (ms)     upload start..end (ms)   upload under a kernel")
    for i, (ks, ke) in enumerate(kern[-10:]):
        # the upload that ends nearest before the NEXT kernel's start belongs to the next step
        ups = [c for c in big if c[0] < ke and c[1] > ks]
        u = ups[0] if ups else None
        under = 0.0
        if u:
            ov = sum(max(0, min(u[1], e) - max(u[0], s)) for s, e in kern)
            under = ov / (u[1] - u[0])
        print(f"{i:3d}   {(ks - t0) / 1e6:8.3f} .. {(ke - t0) / 1e6:8.3f} ({(ke - ks) / 1e6:.3f})   "
              + (f"{(u[0] - t0) / 1e6:8.3f} .. {(u[1] - t0) / 1e6:8.3f} ({(u[1] - u[0]) / 1e6:.3f})   {under:5.2f}" if u else "-"))
    gaps = [kern[i + 1][0] - kern[i][1] for i in range(len(kern) - 10, len(kern) - 1)]
    print(f"idle between consecutive launches (last nine): median {np.median(gaps) / 1e3:.1f} us, max {max(gaps) / 1e3:.1f} us")


if __name__ == "__main__":
    if "--parse" in sys.argv:
        parse(sys.argv[sys.argv.index("--parse") + 1])
    else:
        run()
