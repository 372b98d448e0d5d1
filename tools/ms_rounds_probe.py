#!/usr/bin/env python3
"""Where does k_mean_shift_fused spend its time?  Runs it with 0, 1, 2, 4, 6 rounds (100 launches each, in that order) on
the app's 424x240 label map; under `rocprofv3 --kernel-trace` the per-dispatch durations give the cost of the list
build (0 rounds), of the centroid round and of a weighted round.
    rocprofv3 --kernel-trace --output-format csv -d OUT -- python3 tools/ms_rounds_probe.py ; python3 tools/ms_rounds_probe.py --parse OUT"""
import csv
import glob
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
ROUNDS = [0, 1, 2, 4, 6]
N = 100


def run():
    import torch
    from test_mean_shift import _label_map
    rdf = importlib.import_module("3d-beats_amd")
    msmod = importlib.import_module("3d-beats_amd.cuda.mean_shift")
    lab = _label_map(11, 240, 424, 6, absent=())
    var = np.full(6, 10.0, np.float32)
    dl, dv = rdf.to_device(lab[None]), rdf.to_device(var)
    ms = msmod.MeanShift()
    for r in ROUNDS:
        for _ in range(N):
            ms.run_device(r, dl, 6, dv)
        torch.cuda.synchronize()


def parse(root):
    rows = []
    for f in glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_mean_shift_fused" in r["Kernel_Name"]:
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    rows.sort()
    assert len(rows) == N * len(ROUNDS), len(rows)
    for i, r in enumerate(ROUNDS):
        d = sorted(x[1] for x in rows[i * N:(i + 1) * N])
        print(f"rounds {r}: median {d[len(d) // 2] / 1000:.2f} us  min {d[0] / 1000:.2f} us")


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--parse":
        parse(sys.argv[2])
    else:
        run()
