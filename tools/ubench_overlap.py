#!/usr/bin/env python3
"""Can a kernel with RCCL's footprint start while the forest kernel is running?  

The forest kernel keeps persistent workgroups on every CU; RCCL's send/recv kernel (rcclGenericKernel in librccl.so
for gfx950: 256 threads, ~280 VGPRs, 19.7 KB LDS) needs the register files of a whole CU for one workgroup.  This
launches the forest kernel on stream A and, 1 ms later, 16 workgroups of a stand-in with that footprint on stream B,
and reports when the stand-in finished relative to the forest launch -- with A an ordinary stream, and with A created
by rdf_stream_create_with_reserved_cus (the first n CUs of the mask numbering left to other streams)."""
import ctypes
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    rdf = importlib.import_module("3d-beats_amd")
    rt = rdf.get_runtime()
    lib = rt.lib
    F, H, W = 128, 480, 848
    forest = rdf.DecisionForest.from_numpy(rdf.synth.forest(4, 20, 4, "full"))
    depth = rdf.to_device(rdf.synth.mixed_batch(F))
    labels = rdf.DeviceArray((F, H, W), np.uint16).fill(65535)
    ev = rdf.DecisionTreeEvaluator()
    forest.packed(1.0)
    t_start = rdf.DeviceArray((64,), np.uint64).fill(0)
    side = torch.cuda.Stream()
    spin_ticks = 50_000          # 0.5 ms at 100 MHz
    out = {}
    for name, nth in (("plain stream", 0), ("16 CUs reserved (half of the shader engines)", 16),
                      ("32 CUs reserved (one per shader engine)", 32), ("64 CUs reserved", 64)):
        if nth:
            h = ctypes.c_void_p()
            assert lib.rdf_stream_create_with_reserved_cus(ctypes.byref(h), nth) == 0
            main_stream = torch.cuda.ExternalStream(h.value)
        else:
            h, main_stream = None, torch.cuda.Stream()
        with torch.cuda.stream(main_stream):
            for _ in range(3):
                ev.get_labels_forest(forest, depth, labels)
        with torch.cuda.stream(side):
            assert lib.rdf_debug_fat_kernel(16, 100, t_start.ptr, rt.stream()) == 0
        torch.cuda.synchronize()
        res = []
        for rep in range(5):
            e0, e1, ef = (torch.cuda.Event(enable_timing=True) for _ in range(3))
            with torch.cuda.stream(main_stream):
                e0.record()
                ev.get_labels_forest(forest, depth, labels)
                e1.record()
            time.sleep(0.001)                      # the forest launch is now in flight (it takes ~6 ms)
            with torch.cuda.stream(side):
                assert lib.rdf_debug_fat_kernel(16, spin_ticks, t_start.ptr, rt.stream()) == 0
                ef.record()
            torch.cuda.synchronize()
            res.append((e0.elapsed_time(e1), e0.elapsed_time(ef)))
        forest_ms = float(np.median([r[0] for r in res]))
        fat_done_ms = float(np.median([r[1] for r in res]))
        out[name] = {"forest_kernel_ms": round(forest_ms, 3), "fat_kernel_done_ms_after_forest_start": round(fat_done_ms, 3),
                     "fat_kernel_ran_during_forest": bool(fat_done_ms < forest_ms - 0.2)}
        if h is not None:
            torch.cuda.synchronize()
            lib.rdf_stream_destroy(h)
    # ---- split steps (round 6): the gather of step s-1 occupies the 32 reserved CUs for the FIRST part of step s, then they
    # idle.  Main launch on the masked stream at once; the stand-in (RCCL's footprint, spinning for `ms`) on a side stream at
    # the same time; a helper launch that waits for the stand-in and then pulls tiles from the main launch's queue
    # (rdf_eval_forest_packed_split).  Compared with the masked stream alone and with all 256 CUs and no stand-in at all.
    h = ctypes.c_void_p()
    assert lib.rdf_stream_create_with_reserved_cus(ctypes.byref(h), 32) == 0
    main_stream = torch.cuda.ExternalStream(h.value)
    helper = torch.cuda.Stream()
    plain = torch.cuda.Stream()

    def timed(fn, reps=7):
        res = []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            fn(e0, e1)
            torch.cuda.synchronize()
            res.append(e0.elapsed_time(e1))
        return round(float(np.median(res[2:])), 3)

    def alone(e0, e1):
        with torch.cuda.stream(plain):
            e0.record()
            ev.get_labels_forest(forest, depth, labels)
            e1.record()

    split = {"all 256 CUs, nothing beside it": timed(alone)}
    for ms in (0.5, 1.0, 2.0, 3.0):
        ticks = int(ms * 100_000)

        def masked_only(e0, e1, ticks=ticks):
            with torch.cuda.stream(side):
                assert lib.rdf_debug_fat_kernel(16, ticks, t_start.ptr, rt.stream()) == 0
            with torch.cuda.stream(main_stream):
                e0.record()
                ev.get_labels_forest(forest, depth, labels)
                e1.record()

        def split_step(e0, e1, ticks=ticks):
            with torch.cuda.stream(side):
                assert lib.rdf_debug_fat_kernel(16, ticks, t_start.ptr, rt.stream()) == 0
            helper.wait_stream(side)            # the helper starts when the stand-in has left the reserved CUs
            with torch.cuda.stream(main_stream):
                e0.record()
                ev.get_labels_forest_split(forest, depth, labels, helper.cuda_stream, 32)
                main_stream.wait_stream(helper)
                e1.record()

        split[f"stand-in for {ms} ms"] = {"masked stream alone (224 CUs)": timed(masked_only), "split step (224 CUs + helper on 32)": timed(split_step)}
    out["split step: forest step (ms) next to an RCCL-sized stand-in that holds the 32 reserved CUs from the start of the step"] = split
    torch.cuda.synchronize()
    lib.rdf_stream_destroy(h)

    # the peer-copy gather's data movement: a 104-MB device-to-device hipMemcpyAsync on a side stream, 1 ms into a
    # forest launch on an ordinary stream (on this one-GPU box the copy stays on the device; between GPUs it is the
    # copy engines' job)
    import ctypes as ct
    nbytes = F * H * W * 2
    src = rdf.DeviceArray((nbytes,), np.uint8).fill(1)
    dst = rdf.DeviceArray((nbytes,), np.uint8).fill(0)
    main_stream = torch.cuda.Stream()
    res = []
    for rep in range(6):
        e0, e1, c0, c1 = (torch.cuda.Event(enable_timing=True) for _ in range(4))
        with torch.cuda.stream(main_stream):
            e0.record()
            if rep >= 1:
                ev.get_labels_forest(forest, depth, labels)
            e1.record()
        time.sleep(0.001)
        with torch.cuda.stream(side):
            c0.record()
            assert lib.rdf_memcpy_device_async(ct.c_void_p(dst.ptr), ct.c_void_p(src.ptr), nbytes, rt.stream()) == 0
            c1.record()
        torch.cuda.synchronize()
        res.append((e0.elapsed_time(e1), c0.elapsed_time(c1), e0.elapsed_time(c1)))
    out["104 MB device copy on a side stream"] = {
        "copy_alone_ms": round(res[0][1], 3),
        "forest_kernel_ms_with_copy": round(float(np.median([r[0] for r in res[1:]])), 3),
        "copy_ms_during_forest": round(float(np.median([r[1] for r in res[1:]])), 3),
        "copy_done_ms_after_forest_start": round(float(np.median([r[2] for r in res[1:]])), 3)}
    print(json.dumps({"overlap": out, "fat_kernel": "16 workgroups x 256 threads, 251 VGPRs, 19744 B LDS, spins 0.5 ms; "
                      "launched 1 ms after the forest kernel"}))


if __name__ == "__main__":
    main()
