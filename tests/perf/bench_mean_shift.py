#!/usr/bin/env python3
"""Times rdf_mean_shift on the app's label-map size and against the numpy restatement."""
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch
    from oracle import mean_shift_numpy as ms_np
    from test_mean_shift import _label_map
    rdf = importlib.import_module("3d-beats_amd")
    msmod = importlib.import_module("3d-beats_amd.cuda.mean_shift")
    out = {}
    for (h, w) in [(240, 424), (480, 848)]:
        L, rounds = 6, 6       # 3d_bz.py:65, 108-113
        lab = _label_map(11, h, w, L, absent=())
        var = np.full(L, 10.0, np.float32)
        dl, dv = rdf.to_device(lab[None]), rdf.to_device(var)
        ms = msmod.MeanShift()
        for _ in range(5):
            ms.run_device(rounds, dl, L, dv)
        torch.cuda.synchronize()
        n = 200
        t0 = time.perf_counter()
        for _ in range(n):
            ms.run_device(rounds, dl, L, dv)
        torch.cuda.synchronize()
        dev = (time.perf_counter() - t0) / n
        t0 = time.perf_counter()
        for _ in range(50):
            ms.run(rounds, dl, L, dv)          # + the D2H of the 6x2 means the reference API returns
        host = (time.perf_counter() - t0) / 50
        t0 = time.perf_counter()
        want = ms_np.mean_shift(lab, L, var, rounds)
        cpu = time.perf_counter() - t0
        got = ms.run(rounds, dl, L, dv)
        out[f"{w}x{h}"] = {"device_us_per_run": round(dev * 1e6, 1), "with_result_copy_us": round(host * 1e6, 1),
                           "numpy_restatement_ms": round(cpu * 1e3, 2), "max_abs_diff_px": float(np.nanmax(np.abs(got - want))),
                           "rounds": rounds, "classes": L, "launches": 1,
                           "label_bytes_per_round": h * w * 2}
    print(json.dumps({"mean_shift": out}))


if __name__ == "__main__":
    main()
