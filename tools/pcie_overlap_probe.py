#!/usr/bin/env python3
"""What overlaps with what on this box: pinned H2D of the bench batch's frames, D2H of its labels and the forest kernel, alone
and in pairs on separate streams (wall time per repetition, 5 repetitions after a warm-up).
usage: tools/pcie_overlap_probe.py [--frames 128] | --pipelines | --host-labels"""
import argparse
import os
import sys
import time
from importlib import import_module

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=128)
    a = ap.parse_args()
    import torch
    rdf = import_module("3d-beats_amd")
    synth = rdf.synth
    F, H, W = a.frames, 480, 848
    frames = synth.frames(["dense", "live"] * (F // 2), 0)
    forest = rdf.DecisionForest.from_numpy(synth.forest(4, 20, 4, "full"))
    depth = rdf.to_device(frames)
    labels = rdf.DeviceArray(frames.shape, np.uint16).fill(65535)
    ev = rdf.DecisionTreeEvaluator(use_packed=True)
    pin_in = torch.from_numpy(np.array(frames).view(np.int16).reshape(-1)).pin_memory()
    pin_out = torch.empty(F * H * W, dtype=torch.int16).pin_memory()
    d_t, l_t = depth.torch_bytes().view(torch.int16), labels.torch_bytes().view(torch.int16)
    d2 = torch.empty_like(d_t)          # a second device buffer, so that an upload never races the kernel's reads
    s_up, s_dn = torch.cuda.Stream(), torch.cuda.Stream()
    cur = torch.cuda.current_stream()

    def up():
        with torch.cuda.stream(s_up):
            d2.copy_(pin_in, non_blocking=True)

    def dn():
        with torch.cuda.stream(s_dn):
            pin_out.copy_(l_t, non_blocking=True)

    def kern():
        ev.get_labels_forest(forest, depth, labels)

    def timed(name, *fns):
        for rep in range(6):
            if rep == 1:
                torch.cuda.synchronize()
                t = time.perf_counter()
            for f in fns:
                f()
            # one repetition at a time: every stream drains before the next starts
            torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / 5
        print(f"{name:28s} {dt * 1e3:8.3f} ms")
        return dt

    mb = F * H * W * 2 / 1e6
    print(f"{F} frames: {mb:.0f} MB up, {mb:.0f} MB down; HSA_ENABLE_SDMA={os.environ.get('HSA_ENABLE_SDMA')}")
    t_up = timed("H2D alone", up)
    t_dn = timed("D2H alone", dn)
    t_k = timed("kernel alone", kern)
    timed("H2D || D2H", up, dn)
    timed("H2D || kernel", up, kern)
    timed("kernel || H2D (kernel first)", kern, up)
    timed("D2H || kernel", dn, kern)
    timed("H2D || D2H || kernel", up, dn, kern)
    print(f"H2D {mb / t_up / 1e3:.1f} GB/s, D2H {mb / t_dn / 1e3:.1f} GB/s, kernel {t_k * 1e3:.2f} ms")


def pipelines():
    """Two pipelines over the same work: (i) bench.py's round-3 leg -- four chunks, one device buffer, upload / evaluate / download
    of neighbouring chunks overlap; (ii) whole batches, two device buffers each way: upload of batch r+1, kernel of batch r and
    download of batch r-1 run together."""
    import torch
    rdf = import_module("3d-beats_amd")
    synth = rdf.synth
    F, H, W = 128, 480, 848
    frames = synth.frames(["dense", "live"] * (F // 2), 0)
    forest = rdf.DecisionForest.from_numpy(synth.forest(4, 20, 4, "full"))
    ev = rdf.DecisionTreeEvaluator(use_packed=True)
    pin_in = torch.from_numpy(np.array(frames).view(np.int16).reshape(-1)).pin_memory()
    pin_out = torch.empty(F * H * W, dtype=torch.int16).pin_memory()
    s_up, s_dn = torch.cuda.Stream(), torch.cuda.Stream()
    cur = torch.cuda.current_stream()
    depth = [rdf.DeviceArray(frames.shape, np.uint16) for _ in range(2)]
    labels = [rdf.DeviceArray(frames.shape, np.uint16).fill(65535) for _ in range(2)]
    d_t = [d.torch_bytes().view(torch.int16) for d in depth]
    l_t = [l.torch_bytes().view(torch.int16) for l in labels]
    for n_ch in (1, 2, 4):
        cuts = [(F * c) // n_ch for c in range(n_ch + 1)]
        per = H * W
        up_done = {}
        k_done = {}
        dn_done = {}
        reps = 8
        for rep in range(reps + 2):
            if rep == 2:
                torch.cuda.synchronize()
                tp = time.perf_counter()
            b = rep & 1
            for c in range(n_ch):
                a0, a1 = cuts[c], cuts[c + 1]
                with torch.cuda.stream(s_up):
                    if (rep - 2, c) in k_done:              # the kernel that read this slot two steps ago
                        s_up.wait_event(k_done[(rep - 2, c)])
                    d_t[b][a0 * per:a1 * per].copy_(pin_in[a0 * per:a1 * per], non_blocking=True)
                    e = torch.cuda.Event()
                    e.record(s_up)
                    up_done[(rep, c)] = e
                cur.wait_event(up_done[(rep, c)])
                if (rep - 2, c) in dn_done:                 # this label slot's previous contents are on the host
                    cur.wait_event(dn_done[(rep - 2, c)])
                ev.get_labels_forest(forest, depth[b][a0:a1], labels[b][a0:a1])
                e = torch.cuda.Event()
                e.record(cur)
                k_done[(rep, c)] = e
                with torch.cuda.stream(s_dn):
                    s_dn.wait_event(e)
                    pin_out[a0 * per:a1 * per].copy_(l_t[b][a0 * per:a1 * per], non_blocking=True)
                    e2 = torch.cuda.Event()
                    e2.record(s_dn)
                    dn_done[(rep, c)] = e2
        torch.cuda.synchronize()
        dt = (time.perf_counter() - tp) / reps
        print(f"two device buffers each way, {n_ch} chunk(s) per step: {dt * 1e3:7.3f} ms per step, {F * H * W / dt / 1e6:8.1f} Mpix/s")


def host_labels():
    """The kernel writing its labels straight into pinned, device-mapped host memory: its time next to labels in HBM, and the
    steady-state step of HostFramesEvaluator both ways (labels by the kernel's own stores / downloaded by a copy engine)."""
    import torch
    rdf = import_module("3d-beats_amd")
    dev = import_module("3d-beats_amd.device")
    synth = rdf.synth
    F, H, W = 128, 480, 848
    frames = synth.frames(["dense", "live"] * (F // 2), 0)
    forest = rdf.DecisionForest.from_numpy(synth.forest(4, 20, 4, "full"))
    ev = rdf.DecisionTreeEvaluator(use_packed=True)
    depth = rdf.to_device(frames)
    in_hbm = rdf.DeviceArray(frames.shape, np.uint16)
    in_host, host_view = dev.host_mapped_array(frames.shape, np.uint16)

    def t_kernel(lab, n=6):
        for i in range(n + 1):
            if i == 1:
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            ev.get_labels_forest_filled(forest, depth, lab)
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) / n
    print(f"kernel, labels in HBM      : {t_kernel(in_hbm):.3f} ms")
    print(f"kernel, labels in host mem : {t_kernel(in_host):.3f} ms   (equal: {np.array_equal(in_hbm.get(), host_view)})")
    for by_copy, pieces in ((False, 1), (False, 2), (True, 1), (True, 2), (True, 3)):
        hp = rdf.HostFramesEvaluator(forest, (F, H, W), evaluator=ev, pieces=pieces, labels_by_copy_engine=by_copy)
        hp.mark_steps = True
        for b in range(2):
            hp.frames[b][:] = frames
        for _ in range(24):
            hp.next_frames()
            last = hp.submit()
        hp.result(last)
        torch.cuda.synchronize()
        m = hp.step_marks
        gaps = [m[i].elapsed_time(m[i + 1]) for i in range(3, len(m) - 1)]
        print(f"labels {'by a copy engine' if by_copy else 'by the kernel   '}, {pieces} piece(s): median {np.median(gaps):.3f} mean {np.mean(gaps):.3f} "
              f"max {max(gaps):.2f} ms per step over {len(gaps)} steps")
        del hp
        import gc
        gc.collect()
        torch.cuda.synchronize()


if __name__ == "__main__":
    if "--pipelines" in sys.argv:
        pipelines()
    elif "--host-labels" in sys.argv:
        host_labels()
    else:
        main()
